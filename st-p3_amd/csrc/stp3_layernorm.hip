// stp3_layernorm.hip -- LayerNorm over the channels of every pixel (+ GELU) for the prediction stage on gfx950.
//
// Reference: stp3/layers/convolutions.py:283-307 (`LayerNorm`, channels_last and channels_first forms -- the same
// normalisation over C), used by the ConvNeXt `Block` (:309-345: dwconv -> LayerNorm -> Linear -> GELU -> Linear) and three
// times per `Bottleblock` (:347-380: conv -> LayerNorm -> GELU) inside the Dual_GRU's trust gate.  torch runs it in
// float32 under autocast (a cast in, a cast out, 250 us forward and 390 us backward per call on 28 x 200 x 200 x 64, the
// GELU as two more passes); here one pass forward and one backward over channels-last rows in the tensor's own type,
// float32 arithmetic, one rounding at the end.  HBM-bound: 2 x (read + write) of the tensor per training step and layer.
//
// A row (pixel) is C contiguous channels; L = C / VEC lanes of a wavefront hold it (16 bytes each) and reduce with
// xor-shuffles, so a 64-channel bf16 row is 8 lanes and a wavefront normalises 8 pixels per step.  C / VEC must be a power
// of two in 4 .. 64 (C = 32 .. 512 in bf16, 16 .. 256 in float32): the prediction stage has 32 and 64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

struct LnDims {
    int64_t rows;
    int C, ldx, ldy, act;
    float eps;
};

constexpr int kThreads = 256;
constexpr int kMaxBwdBlocks = 1024;          // partial rows of the parameter gradients (one resident round of 4 per CU)

template <typename T> struct Row;
template <> struct Row<float> {
    static constexpr int VEC = 4;
    static __device__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ void store(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Row<uint16_t> {
    static constexpr int VEC = 8;
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

// sum over the L lanes that hold one row (L a power of two, the lanes are consecutive)
template <int L>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// exact GELU (nn.GELU() default): 0.5 x (1 + erf(x / sqrt 2)) and its derivative
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float v) {
    return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * expf(-0.5f * v * v);
}

// mean and 1 / sqrt(var + eps) of the row this lane group holds (two passes over the registers, like torch's float32 path)
template <int VEC, int L>
__device__ __forceinline__ void row_stats(const float* v, int C, float eps, float* dev, float* rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += v[j];
    const float mean = row_sum<L>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        dev[j] = v[j] - mean;
        q = fmaf(dev[j], dev[j], q);
    }
    *rstd = 1.0f / sqrtf(row_sum<L>(q) / (float)C + eps);
}

template <typename T, int L, bool GELU>
__global__ __launch_bounds__(kThreads) void layernorm_fwd_kernel(LnDims d, const T* __restrict__ x,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ y) {
    constexpr int VEC = Row<T>::VEC;
    constexpr int RPB = kThreads / L;                       // rows per workgroup and step
    const int lane = threadIdx.x % L, rl = threadIdx.x / L;
    const int c0 = lane * VEC;
    float ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        ga[j] = gamma ? gamma[c0 + j] : 1.f;
        be[j] = beta ? beta[c0 + j] : 0.f;
    }
    // (the trip count is the workgroup's, not the row's: every lane of a wavefront takes part in the shuffles)
    for (int64_t base = (int64_t)blockIdx.x * RPB; base < d.rows; base += (int64_t)gridDim.x * RPB) {
        const int64_t r = base + rl;
        const bool live = r < d.rows;
        float v[VEC], dev[VEC], rstd;
        if (live) {
            Row<T>::load(x + r * d.ldx + c0, v);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = 0.f;
        }
        row_stats<VEC, L>(v, d.C, d.eps, dev, &rstd);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float o = fmaf(dev[j] * rstd, ga[j], be[j]);
            v[j] = GELU ? gelu(o) : o;
        }
        if (live) Row<T>::store(y + r * d.ldy + c0, v);
    }
}

// dx = rstd (g gamma - mean_c(g gamma) - xhat mean_c(g gamma xhat)),  g = dy gelu'(xhat gamma + beta)  (g = dy without the
// activation);  partial[block][0][c] = sum over the block's rows of g xhat (dgamma), [1][c] of g (dbeta)
template <typename T, int L, bool GELU>
__global__ __launch_bounds__(kThreads) void layernorm_bwd_kernel(LnDims d, const T* __restrict__ dy,
                                                                 const T* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ dx,
                                                                 float* __restrict__ partial) {
    constexpr int VEC = Row<T>::VEC;
    constexpr int RPB = kThreads / L;
    __shared__ float red[2][kThreads * VEC];
    const int lane = threadIdx.x % L, rl = threadIdx.x / L;
    const int c0 = lane * VEC;
    float ga[VEC], be[VEC], dga[VEC], dbe[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        ga[j] = gamma ? gamma[c0 + j] : 1.f;
        be[j] = beta ? beta[c0 + j] : 0.f;
        dga[j] = dbe[j] = 0.f;
    }
    const float inv_c = 1.f / (float)d.C;
    for (int64_t base = (int64_t)blockIdx.x * RPB; base < d.rows; base += (int64_t)gridDim.x * RPB) {
        const int64_t r = base + rl;
        const bool live = r < d.rows;
        float v[VEC], g[VEC], dev[VEC], rstd;
        if (live) {
            Row<T>::load(x + r * d.ldx + c0, v);
            Row<T>::load(dy + r * d.ldy + c0, g);
        } else {                                              // a row past the end: zero gradient, nothing stored
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = g[j] = 0.f;
        }
        row_stats<VEC, L>(v, d.C, d.eps, dev, &rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = dev[j] * rstd;
            if (GELU) g[j] *= gelu_grad(fmaf(xh, ga[j], be[j]));
            dga[j] = fmaf(g[j], xh, dga[j]);
            dbe[j] += g[j];
            dev[j] = xh;
            g[j] *= ga[j];
            s1 += g[j];
            s2 = fmaf(g[j], xh, s2);
        }
        s1 = row_sum<L>(s1) * inv_c;
        s2 = row_sum<L>(s2) * inv_c;
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = rstd * (g[j] - s1 - dev[j] * s2);
        if (live) Row<T>::store(dx + r * d.ldx + c0, v);
    }
    // the workgroup's RPB row lanes, summed in a fixed order
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        red[0][rl * d.C + c0 + j] = dga[j];
        red[1][rl * d.C + c0 + j] = dbe[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d.C; i += kThreads) {
        const int k = i / d.C, c = i % d.C;
        float s = 0.f;
        for (int q = 0; q < RPB; ++q) s += red[k][q * d.C + c];
        partial[((size_t)blockIdx.x * 2 + k) * d.C + c] = s;
    }
}

// out[k][c] = sum over the blocks' partial rows, in double, fixed order (k = 0: dgamma, 1: dbeta)
__global__ __launch_bounds__(kThreads) void layernorm_reduce_kernel(int parts, int width, const float* __restrict__ partial,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[kThreads];
    const int il = threadIdx.x % 8, pl = threadIdx.x / 8;
    const int i = blockIdx.x * 8 + il;
    double s0 = 0.0, s1 = 0.0;
    if (i < width) {
        int p = pl;
        for (; p + 32 < parts; p += 64) {
            s0 += (double)partial[(size_t)p * width + i];
            s1 += (double)partial[(size_t)(p + 32) * width + i];
        }
        for (; p < parts; p += 32) s0 += (double)partial[(size_t)p * width + i];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if (pl < st) red[threadIdx.x] += red[threadIdx.x + st * 8];
        __syncthreads();
    }
    if (pl == 0 && i < width) {
        const int c = width / 2;
        if (i < c) { if (dgamma) dgamma[i] = (float)red[il]; }
        else if (dbeta) dbeta[i - c] = (float)red[il];
    }
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// lanes per row, or 0 when the shape is not one the kernels take
inline int check(const stp3_layernorm_dims* p, LnDims* d) {
    if (!p || p->rows <= 0 || p->C <= 0 || p->ldx < p->C || p->ldy < p->C) return -1;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return 0;
    if (p->act != STP3_ACT_NONE && p->act != STP3_ACT_GELU) return -1;
    const int vec = p->dtype == STP3_DTYPE_BF16 ? 8 : 4;
    if (p->C % vec || p->ldx % vec || p->ldy % vec) return 0;
    const int L = p->C / vec;
    if (L < 4 || L > 64 || (L & (L - 1))) return 0;
    d->rows = p->rows; d->C = p->C; d->ldx = p->ldx; d->ldy = p->ldy; d->act = p->act; d->eps = p->eps;
    return L;
}

// run CALL with `LL` (lanes per row) and `GELU` bound as compile-time constants
#define LN_LANES_G(L, G, ...)                                                      \
    switch (L) {                                                                   \
        case 4:  { constexpr int LL = 4;  constexpr bool GELU = G; __VA_ARGS__; break; }  \
        case 8:  { constexpr int LL = 8;  constexpr bool GELU = G; __VA_ARGS__; break; }  \
        case 16: { constexpr int LL = 16; constexpr bool GELU = G; __VA_ARGS__; break; }  \
        case 32: { constexpr int LL = 32; constexpr bool GELU = G; __VA_ARGS__; break; }  \
        default: { constexpr int LL = 64; constexpr bool GELU = G; __VA_ARGS__; break; }  \
    }
#define LN_LANES(L, ...)                                                           \
    do {                                                                           \
        if (d.act == STP3_ACT_GELU) LN_LANES_G(L, true, __VA_ARGS__)               \
        else LN_LANES_G(L, false, __VA_ARGS__)                                     \
    } while (0)

inline int blocks_for(int64_t rows, int L, int cap) {
    const int rpb = kThreads / L;
    const int64_t want = (rows + rpb - 1) / rpb;
    return (int)(want < cap ? want : cap);
}

}  // namespace

extern "C" {

int stp3_layernorm_fwd(const stp3_layernorm_dims* p, const void* x, const float* gamma, const float* beta, void* y,
                       void* stream) {
    LnDims d;
    const int L = check(p, &d);
    if (L < 0) return STP3_EINVAL;
    if (L == 0) return STP3_EUNSUP;
    if (!x || !y) return STP3_EINVAL;
    if (!aligned16(x) || !aligned16(y)) return STP3_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    const int bx = blocks_for(d.rows, L, 256 * 8 * 4);                  // a few resident rounds: every row is one step
    if (p->dtype == STP3_DTYPE_BF16) {
        LN_LANES(L, hipLaunchKernelGGL((layernorm_fwd_kernel<uint16_t, LL, GELU>), dim3(bx), dim3(kThreads), 0, s, d,
                                       (const uint16_t*)x, gamma, beta, (uint16_t*)y));
    } else {
        LN_LANES(L, hipLaunchKernelGGL((layernorm_fwd_kernel<float, LL, GELU>), dim3(bx), dim3(kThreads), 0, s, d, (const float*)x,
                                       gamma, beta, (float*)y));
    }
    return status();
}

int stp3_layernorm_bwd_workspace(const stp3_layernorm_dims* p, size_t* bytes) {
    LnDims d;
    const int L = check(p, &d);
    if (L < 0 || !bytes) return STP3_EINVAL;
    if (L == 0) return STP3_EUNSUP;
    *bytes = (size_t)kMaxBwdBlocks * 2 * p->C * sizeof(float);
    return STP3_OK;
}

int stp3_layernorm_bwd(const stp3_layernorm_dims* p, const void* dy, const void* x, const float* gamma, const float* beta,
                       void* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    LnDims d;
    const int L = check(p, &d);
    if (L < 0) return STP3_EINVAL;
    if (L == 0) return STP3_EUNSUP;
    if (!dy || !x || !dx || !workspace) return STP3_EINVAL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return STP3_EUNSUP;
    if (workspace_bytes < (size_t)kMaxBwdBlocks * 2 * p->C * sizeof(float)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int bx = blocks_for(d.rows, L, kMaxBwdBlocks);
    float* partial = (float*)workspace;
    if (p->dtype == STP3_DTYPE_BF16) {
        LN_LANES(L, hipLaunchKernelGGL((layernorm_bwd_kernel<uint16_t, LL, GELU>), dim3(bx), dim3(kThreads), 0, s, d,
                                       (const uint16_t*)dy, (const uint16_t*)x, gamma, beta, (uint16_t*)dx, partial));
    } else {
        LN_LANES(L, hipLaunchKernelGGL((layernorm_bwd_kernel<float, LL, GELU>), dim3(bx), dim3(kThreads), 0, s, d, (const float*)dy,
                                       (const float*)x, gamma, beta, (float*)dx, partial));
    }
    if (dgamma || dbeta)
        hipLaunchKernelGGL(layernorm_reduce_kernel, dim3((2 * d.C + 7) / 8), dim3(kThreads), 0, s, bx, 2 * d.C,
                           (const float*)partial, dgamma, dbeta);
    return status();
}

}  // extern "C"
