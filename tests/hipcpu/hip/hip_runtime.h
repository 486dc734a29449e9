// TEST INFRASTRUCTURE -- a host-side stand-in for <hip/hip_runtime.h> that lets the repository's .hip sources be
// compiled for x86 (clang++) and EXECUTED on CPU threads, one OS thread per HIP thread, one workgroup at a time.
// Purpose: this build container has no GPU; with this shim the real kernel source -- index arithmetic, LDS layouts,
// barriers, wave-level intrinsics (DPP, readlane, ballot, shuffles) and the MFMA lane maps -- runs under the same C ABI
// (tests/hipcpu/build.py -> libstp3hip_cpu.so) and is compared with the oracle on small problems.  It checks the
// LOGIC of a kernel, not its performance, and it cannot see hardware-only effects (memory model, occupancy limits).
//
// Semantics implemented (what the kernels in st-p3_amd/csrc use, nothing more):
//   * every HIP thread is a FIBER (ucontext); the fibers of a workgroup are scheduled round-robin on one OS thread,
//     several workgroups run in parallel on a small pool of OS threads (shared memory is thread_local to the pool
//     thread).  __syncthreads() is a barrier over the threads of the block that have not returned yet (a returned
//     thread drops out, as on the GPU)
//   * a wave = 64 consecutive threads; wave intrinsics exchange values through a per-wave buffer with two wave
//     barriers, so they must be reached by all live lanes of the wave (convergent use) -- if no fiber of a block can
//     make progress the run aborts with a message (exact deadlock detection, no timeouts)
//   * v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x32_bf16 with the operand / result lane maps of the CDNA3/4 ISA
//     (cdna_hip_programming.md section 3): A[i = l & 15][k-chunk = l >> 4], B[k-chunk = l >> 4][j = l & 15],
//     D[4 * (l >> 4) + r][l & 15]; fp32 products are chained with fmaf in ascending k
//   * __shared__ arrays are `static thread_local` (one block per pool thread at a time); `extern __shared__`
//     declarations are rewritten by the build script into a pointer to a 160 KB thread_local buffer
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define warpSize 64

typedef void* hipStream_t;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 8; return hipSuccess; }   // "8 CUs": two 4-wave workgroups per XCD slot
template <typename F>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }
// launches are synchronous here, so an asynchronous memset is a memset
inline hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) { std::memset(p, value, bytes); return hipSuccess; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }

namespace hipcpu {

constexpr int kWave = 64;
constexpr size_t kDynLds = 160 * 1024;

[[noreturn]] inline void die(const char* what) {
    std::fprintf(stderr, "hipcpu: %s\n", what);
    std::fflush(stderr);
    std::abort();
}

void yield_to_scheduler();

// barrier over the fibers that are still alive; drop() removes a participant for good.  All fibers of a block live on
// one OS thread, so plain counters suffice.
struct Barrier {
    int live = 0, arrived = 0;
    unsigned gen = 0;
    void reset(int n) { live = n; arrived = 0; ++gen; }
    void wait();
    void drop() {
        --live;
        if (live > 0 && arrived >= live) { arrived = 0; ++gen; }
    }
};

struct Wave {
    Barrier bar;
    unsigned long long live = 0;                    // lanes whose thread has not returned yet
    alignas(16) unsigned char slot[2][kWave][32];   // two operands, up to 32 bytes per lane
};

struct Block {
    Barrier bar;
    std::vector<Wave> waves;
};

struct Ctx {
    dim3 tid, bid, bdim, gdim;
    int lane = 0, wave = 0;
    Block* block = nullptr;
    const Barrier* waiting_on = nullptr;      // set while parked in a barrier
    unsigned wait_gen = 0;
};
extern thread_local Ctx* cur;                 // the fiber that is running on this OS thread
extern thread_local unsigned char g_dyn_lds[kDynLds];
inline void* dyn_lds() { return g_dyn_lds; }

inline void Barrier::wait() {
    const unsigned g = gen;
    if (++arrived >= live) { arrived = 0; ++gen; return; }
    cur->waiting_on = this;
    cur->wait_gen = g;
    while (gen == g) yield_to_scheduler();
    cur->waiting_on = nullptr;
}

inline Wave& my_wave() { return cur->block->waves[cur->wave]; }

// every live lane of the wave deposits `v`, then reads the value deposited by lane `src`
template <typename T>
inline T exchange(T v, int src) {
    static_assert(sizeof(T) <= 32, "exchange payload");
    Wave& w = my_wave();
    std::memcpy(w.slot[0][cur->lane], &v, sizeof(T));
    w.bar.wait();
    T r;
    std::memcpy(&r, w.slot[0][src & (kWave - 1)], sizeof(T));
    w.bar.wait();
    return r;
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& kernel);

}  // namespace hipcpu

#define threadIdx (hipcpu::cur->tid)
#define blockIdx (hipcpu::cur->bid)
#define blockDim (hipcpu::cur->bdim)
#define gridDim (hipcpu::cur->gdim)
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    hipcpu::launch((grid), (block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { hipcpu::cur->block->bar.wait(); }
inline void __builtin_amdgcn_wave_barrier_() { hipcpu::my_wave().bar.wait(); }
#define __builtin_amdgcn_wave_barrier __builtin_amdgcn_wave_barrier_
#define __builtin_amdgcn_fence(order, scope) std::atomic_thread_fence(std::memory_order_seq_cst)

// ---- bit casts / math ----
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __expf(float x) { return std::exp(x); }
using std::min;
using std::max;
inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }

template <typename T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

// ---- wave intrinsics ----
inline int readlane_i(int v, int lane) { return hipcpu::exchange<int>(v, lane); }
#define __builtin_amdgcn_readlane(v, lane) readlane_i((int)(v), (int)(lane))
// v_readfirstlane_b32: the value of the lowest live lane, for every lane (NOT the lane's own value: a kernel that
// applies it to a non-uniform quantity behaves here as it would on the GPU)
inline int readfirstlane_i(int v) {
    hipcpu::Wave& w = hipcpu::my_wave();
    std::memcpy(w.slot[0][hipcpu::cur->lane], &v, 4);
    w.bar.wait();
    int r;
    std::memcpy(&r, w.slot[0][__builtin_ctzll(w.live)], 4);
    w.bar.wait();
    return r;
}
#define __builtin_amdgcn_readfirstlane(v) readfirstlane_i((int)(v))

template <typename T>
inline T __shfl(T v, int src, int width = 64) { (void)width; return hipcpu::exchange<T>(v, src); }
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hipcpu::exchange<T>(v, hipcpu::cur->lane ^ mask); }
template <typename T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
    (void)width;
    const int src = hipcpu::cur->lane - (int)delta;
    const T r = hipcpu::exchange<T>(v, src < 0 ? hipcpu::cur->lane : src);
    return src < 0 ? v : r;
}
inline unsigned long long __ballot(int pred) {
    hipcpu::Wave& w = hipcpu::my_wave();
    const int p = pred ? 1 : 0;
    std::memcpy(w.slot[0][hipcpu::cur->lane], &p, 4);
    std::memset(w.slot[1][hipcpu::cur->lane], 1, 1);                  // "this lane is alive and voted"
    w.bar.wait();
    unsigned long long m = 0;
    for (int l = 0; l < hipcpu::kWave; ++l) {
        int q = 0;
        std::memcpy(&q, w.slot[0][l], 4);
        if (w.slot[1][l][0] == 1 && q) m |= 1ull << l;
    }
    w.bar.wait();
    w.slot[1][hipcpu::cur->lane][0] = 0;
    return m;
}

inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }

// v_perm_b32: byte select from {a (bytes 7..4), b (bytes 3..0)}; selectors 0..7 pick a byte, 0x0c gives 0x00
inline unsigned __builtin_amdgcn_perm_(unsigned a, unsigned b, unsigned sel) {
    const unsigned long long src = ((unsigned long long)a << 32) | b;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned s = (sel >> (8 * i)) & 0xff;
        unsigned byte = 0;
        if (s < 8) byte = (unsigned)(src >> (8 * s)) & 0xff;
        else if (s == 0x0c) byte = 0x00;
        else if (s >= 0x0d) byte = 0xff;
        else hipcpu::die("v_perm_b32 selector not modelled");
        r |= byte << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_perm __builtin_amdgcn_perm_

// DPP: the controls used here -- quad_perm (0x00..0xff), row_mirror 0x140, row_half_mirror 0x141, row_bcast15 0x142,
// row_bcast31 0x143, row_newbcast:n 0x150+n, row_ror:n 0x120+n; lanes of rows that row_mask disables (or without a valid source) receive `old`
inline int update_dpp_(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)bank_mask; (void)bound_ctrl;
    const int lane = hipcpu::cur->lane, row = lane >> 4;
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x142) from = row >= 1 ? 16 * (row - 1) + 15 : -1;
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;
    else if (ctrl >= 0x150 && ctrl <= 0x15f) from = (lane & ~15) | (ctrl & 15);          // row_newbcast:n (gfx90a+)
    else if (ctrl >= 0x121 && ctrl <= 0x12f) from = (lane & ~15) | ((lane - (ctrl & 15)) & 15);   // row_ror:n
    else hipcpu::die("DPP control not modelled");
    const int got = hipcpu::exchange<int>(src, from < 0 ? lane : from);
    if (from < 0 || !((row_mask >> row) & 1)) return old;
    return got;
}
#define __builtin_amdgcn_update_dpp update_dpp_

// ---- MFMA ----
typedef float hipcpu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipcpu_bf16x8 __attribute__((ext_vector_type(8)));

inline hipcpu_f32x4 mfma_f32_16x16x4f32_(float a, float b, hipcpu_f32x4 c, int, int, int) {
    hipcpu::Wave& w = hipcpu::my_wave();
    const int lane = hipcpu::cur->lane;
    std::memcpy(w.slot[0][lane], &a, 4);
    std::memcpy(w.slot[1][lane], &b, 4);
    w.bar.wait();
    hipcpu_f32x4 d = c;
    const int j = lane & 15;
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * (lane >> 4) + q;
        float acc = c[q];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, w.slot[0][i + 16 * k], 4);
            std::memcpy(&bv, w.slot[1][j + 16 * k], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[q] = acc;
    }
    w.bar.wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 mfma_f32_16x16x4f32_

inline float hipcpu_bf16_to_float(const unsigned char* p) {
    unsigned short h;
    std::memcpy(&h, p, 2);
    return __uint_as_float((unsigned)h << 16);
}

template <typename V>
inline hipcpu_f32x4 mfma_f32_16x16x32_bf16_(V a, V b, hipcpu_f32x4 c, int, int, int) {
    static_assert(sizeof(V) == 16, "8 x bf16 per lane");
    hipcpu::Wave& w = hipcpu::my_wave();
    const int lane = hipcpu::cur->lane;
    std::memcpy(w.slot[0][lane], &a, 16);
    std::memcpy(w.slot[1][lane], &b, 16);
    w.bar.wait();
    hipcpu_f32x4 d = c;
    const int j = lane & 15;
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * (lane >> 4) + q;
        float acc = c[q];
        for (int kk = 0; kk < 4; ++kk)
            for (int t = 0; t < 8; ++t)
                acc += hipcpu_bf16_to_float(w.slot[0][i + 16 * kk] + 2 * t) * hipcpu_bf16_to_float(w.slot[1][j + 16 * kk] + 2 * t);
        d[q] = acc;
    }
    w.bar.wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 mfma_f32_16x16x32_bf16_

// v_mfma_f32_32x32x2_f32: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], D as for every 32x32 MFMA:
// D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31], r = 0..15
typedef float hipcpu_f32x16a __attribute__((ext_vector_type(16)));
inline hipcpu_f32x16a mfma_f32_32x32x2f32_(float a, float b, hipcpu_f32x16a c, int, int, int) {
    hipcpu::Wave& w = hipcpu::my_wave();
    const int lane = hipcpu::cur->lane;
    std::memcpy(w.slot[0][lane], &a, 4);
    std::memcpy(w.slot[1][lane], &b, 4);
    w.bar.wait();
    hipcpu_f32x16a d = c;
    const int j = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, w.slot[0][i + 32 * k], 4);
            std::memcpy(&bv, w.slot[1][j + 32 * k], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    w.bar.wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 mfma_f32_32x32x2f32_

// v_mfma_f32_32x32x16_bf16: A[i = l & 31][k = 8 * (l >> 5) + t], B[k = 8 * (l >> 5) + t][j = l & 31] (t = 0..7),
// D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31], r = 0..15 (cdna_hip_programming.md section 3)
typedef float hipcpu_f32x16 __attribute__((ext_vector_type(16)));
template <typename V>
inline hipcpu_f32x16 mfma_f32_32x32x16_bf16_(V a, V b, hipcpu_f32x16 c, int, int, int) {
    static_assert(sizeof(V) == 16, "8 x bf16 per lane");
    hipcpu::Wave& w = hipcpu::my_wave();
    const int lane = hipcpu::cur->lane;
    std::memcpy(w.slot[0][lane], &a, 16);
    std::memcpy(w.slot[1][lane], &b, 16);
    w.bar.wait();
    hipcpu_f32x16 d = c;
    const int j = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int half = 0; half < 2; ++half)
            for (int t = 0; t < 8; ++t)
                acc += hipcpu_bf16_to_float(w.slot[0][i + 32 * half] + 2 * t) * hipcpu_bf16_to_float(w.slot[1][j + 32 * half] + 2 * t);
        d[r] = acc;
    }
    w.bar.wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 mfma_f32_32x32x16_bf16_
