// stp3_se.hip -- squeeze-and-excitation data passes for gfx950 (channels-last, bf16 / f32):
//   stp3_se_pool  : out[n][c] = sum_r x[n][r][c]                (dy == NULL)   the "squeeze"
//                   out[n][c] = sum_r dy[n][r][c] * x[n][r][c]  (dy != NULL)   gradient of the gate
//   stp3_se_scale : y[n][r][c] = x[n][r][c] * gate[n][c] (+ add[n][c])          the "excite" and its input gradient
// Replaces, inside the EfficientNet MBConv blocks that stp3/models/encoder.py:57-97 drives (efficientnet_pytorch's
// MBConvBlock: adaptive_avg_pool2d -> _se_reduce -> swish -> _se_expand -> sigmoid * x), the pooling, the gate
// multiply and -- in the backward -- the four elementwise / reduction passes autograd derives from them
// (dy*sigmoid, dy*x, its sum, and the add of the two input-gradient terms) by two streaming passes.
// HBM-bound: 16-byte channel vectors, float32 accumulation, deterministic two-stage reduction.
// STATUS: compiled and exported; host side (stp3_amd/ops_fused.py) is selected only with STP3_FUSED_SE=1 and is
// not yet validated on hardware.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

struct SeDims {
    int N, rows, C, ld;
};
constexpr int kT = 256;
constexpr int kMaxBx = 64;

template <typename T, int VEC> struct Io2;
template <> struct Io2<float, 4> {
    static __device__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ void store(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Io2<float, 1> {
    static __device__ void load(const float* p, float* f) { f[0] = p[0]; }
    static __device__ void store(float* p, const float* f) { p[0] = f[0]; }
};
template <> struct Io2<uint16_t, 8> {
    static __device__ void load(const uint16_t* p, float* f) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ void store(uint16_t* p, const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Io2<uint16_t, 1> {
    static __device__ void load(const uint16_t* p, float* f) { f[0] = __uint_as_float((uint32_t)p[0] << 16); }
    static __device__ void store(uint16_t* p, const float* f) { p[0] = (uint16_t)pack_bf16(f[0], 0.f); }
};

// thread -> (row lane, channel vector); grid = (row blocks, N, channel tiles)
template <int VEC>
__device__ __forceinline__ void se_map(const SeDims& d, int& cv, int& rl, int& RL, int& CVB, bool& live) {
    const int CV = (d.C + VEC - 1) / VEC;
    CVB = min(CV, kT);
    RL = 1;
    while (RL * 2 * CVB <= kT) RL *= 2;
    const int cvb = threadIdx.x % CVB;
    rl = threadIdx.x / CVB;
    cv = blockIdx.z * CVB + cvb;
    live = rl < RL && cv < CV;
}

// partial[(n * gridDim.x + bx) * C + c]
template <typename T, int VEC, bool PRODUCT>
__global__ __launch_bounds__(kT) void se_pool_kernel(SeDims d, const T* __restrict__ x, const T* __restrict__ dy,
                                                     float* __restrict__ partial) {
    __shared__ float red[kT * VEC];
    int cv, rl, RL, CVB;
    bool live;
    se_map<VEC>(d, cv, rl, RL, CVB, live);
    const int n = blockIdx.y;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (live) {
        const T* xs = x + (size_t)n * d.rows * d.ld + cv * VEC;
        const T* gs = PRODUCT ? dy + (size_t)n * d.rows * d.ld + cv * VEC : nullptr;
        const int step = gridDim.x * RL;
        for (int r = blockIdx.x * RL + rl; r < d.rows; r += 2 * step) {
            float a[2][VEC], b[2][VEC];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r + u * step < d.rows) {
                    Io2<T, VEC>::load(xs + (size_t)(r + u * step) * d.ld, a[u]);
                    if (PRODUCT) Io2<T, VEC>::load(gs + (size_t)(r + u * step) * d.ld, b[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r + u * step < d.rows) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] = PRODUCT ? fmaf(a[u][j], b[u][j], acc[j]) : acc[j] + a[u][j];
                }
            }
        }
    }
    const int cvb = threadIdx.x % CVB;
    const int width = CVB * VEC;
    if (rl < RL) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[rl * width + cvb * VEC + j] = live ? acc[j] : 0.f;
    }
    __syncthreads();
    for (int s = RL >> 1; s > 0; s >>= 1) {
        if (rl < s) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[rl * width + cvb * VEC + j] += red[(rl + s) * width + cvb * VEC + j];
        }
        __syncthreads();
    }
    if (rl == 0 && live) {
        float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * d.C + cv * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (cv * VEC + j < d.C) out[j] = red[cvb * VEC + j];
    }
}

// out[n][c] = sum_b partial[(n * parts + b) * C + c]: 64 channels x 4 part lanes per workgroup, fixed summation order
__global__ __launch_bounds__(kT) void se_reduce_kernel(int parts, int C, const float* __restrict__ partial,
                                                       float* __restrict__ out) {
    __shared__ float red[kT];
    const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int n = blockIdx.y;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float* src = partial + (size_t)n * parts * C + c;
        int b = bl;
        for (; b + 4 < parts; b += 8) {
            const float a = src[(size_t)b * C], e = src[(size_t)(b + 4) * C];
            s0 += a; s1 += e;
        }
        for (; b < parts; b += 4) s0 += src[(size_t)b * C];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (bl == 0 && c < C) out[(size_t)n * C + c] = (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
}

template <typename T, int VEC>
__global__ __launch_bounds__(kT) void se_scale_kernel(SeDims d, const T* __restrict__ x, const float* __restrict__ gate,
                                                      const float* __restrict__ add, T* __restrict__ y) {
    int cv, rl, RL, CVB;
    bool live;
    se_map<VEC>(d, cv, rl, RL, CVB, live);
    if (!live) return;
    const int n = blockIdx.y;
    const int c0 = cv * VEC;
    float g[VEC], a[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const bool ok = c0 + j < d.C;
        g[j] = ok ? gate[(size_t)n * d.C + c0 + j] : 0.f;
        a[j] = (ok && add) ? add[(size_t)n * d.C + c0 + j] : 0.f;
    }
    const T* xs = x + (size_t)n * d.rows * d.ld + c0;
    T* ys = y + (size_t)n * d.rows * d.ld + c0;
    const int step = gridDim.x * RL;
    for (int r = blockIdx.x * RL + rl; r < d.rows; r += 4 * step) {
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r + u * step < d.rows) Io2<T, VEC>::load(xs + (size_t)(r + u * step) * d.ld, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r + u * step < d.rows) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u][j] = fmaf(v[u][j], g[j], a[j]);
                Io2<T, VEC>::store(ys + (size_t)(r + u * step) * d.ld, v[u]);
            }
        }
    }
}

struct SePlan {
    SeDims d;
    int vec;
    bool bf16;
    dim3 grid;
};

inline int se_plan(const stp3_se_dims* p, SePlan* P, std::initializer_list<const void*> ptrs) {
    if (!p || p->N <= 0 || p->rows <= 0 || p->C <= 0 || p->ld < p->C) return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    P->bf16 = p->dtype == STP3_DTYPE_BF16;
    const int wide = P->bf16 ? 8 : 4;
    bool ok = p->C % wide == 0 && p->ld % wide == 0;
    for (const void* q : ptrs) ok = ok && (q == nullptr || ((uintptr_t)q & 15) == 0);
    P->vec = ok ? wide : 1;
    P->d.N = p->N; P->d.rows = p->rows; P->d.C = p->C; P->d.ld = p->ld;
    const int CV = (p->C + P->vec - 1) / P->vec;
    const int CVB = CV < kT ? CV : kT;
    int RL = 1;
    while (RL * 2 * CVB <= kT) RL *= 2;
    const int ctiles = (CV + CVB - 1) / CVB;
    int bx = (2048 + p->N * ctiles - 1) / (p->N * ctiles);
    const int max_bx = (p->rows + RL * 8 - 1) / (RL * 8);
    if (bx > max_bx) bx = max_bx;
    if (bx < 1) bx = 1;
    if (bx > kMaxBx) bx = kMaxBx;
    P->grid = dim3(bx, p->N, ctiles);
    return STP3_OK;
}

#define SE_SWITCH(P, ...)                                                         \
    do {                                                                          \
        if ((P).bf16) {                                                           \
            if ((P).vec == 8) { using T = uint16_t; constexpr int VEC = 8; __VA_ARGS__; } \
            else              { using T = uint16_t; constexpr int VEC = 1; __VA_ARGS__; } \
        } else {                                                                  \
            if ((P).vec == 4) { using T = float; constexpr int VEC = 4; __VA_ARGS__; }    \
            else              { using T = float; constexpr int VEC = 1; __VA_ARGS__; }    \
        }                                                                         \
    } while (0)

inline int se_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

}  // namespace

extern "C" {

int stp3_se_workspace_bytes(const stp3_se_dims* p, size_t* bytes) {
    if (!p || !bytes || p->N <= 0 || p->C <= 0) return STP3_EINVAL;
    *bytes = (size_t)p->N * kMaxBx * p->C * sizeof(float);
    return STP3_OK;
}

int stp3_se_pool(const stp3_se_dims* p, const void* x, const void* dy, void* workspace, size_t workspace_bytes, float* out,
                 void* stream) {
    SePlan P;
    int rc = se_plan(p, &P, {x, dy});
    if (rc) return rc;
    if (!x || !workspace || !out) return STP3_EINVAL;
    if (workspace_bytes < (size_t)p->N * kMaxBx * p->C * sizeof(float)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (dy)
        SE_SWITCH(P, hipLaunchKernelGGL((se_pool_kernel<T, VEC, true>), P.grid, dim3(kT), 0, s, P.d, (const T*)x, (const T*)dy,
                                        partial));
    else
        SE_SWITCH(P, hipLaunchKernelGGL((se_pool_kernel<T, VEC, false>), P.grid, dim3(kT), 0, s, P.d, (const T*)x,
                                        (const T*)nullptr, partial));
    hipLaunchKernelGGL(se_reduce_kernel, dim3((p->C + 63) / 64, p->N), dim3(kT), 0, s, (int)P.grid.x, p->C, partial, out);
    return se_status();
}

int stp3_se_scale(const stp3_se_dims* p, const void* x, const float* gate, const float* add, void* y, void* stream) {
    SePlan P;
    int rc = se_plan(p, &P, {x, y});
    if (rc) return rc;
    if (!x || !gate || !y) return STP3_EINVAL;
    SE_SWITCH(P, hipLaunchKernelGGL((se_scale_kernel<T, VEC>), P.grid, dim3(kT), 0, (hipStream_t)stream, P.d, (const T*)x, gate,
                                    add, (T*)y));
    return se_status();
}

}  // extern "C"
