// stp3_bnact.hip -- fused BatchNorm (+ per-sample bias) + activation (+ residual) for gfx950,
// channels-last, forward and backward, with the batch statistics exposed between the two passes
// so that the host can all-reduce them across ranks (cross-replica BatchNorm).
//
// Replaces the nn.BatchNorm2d/3d -> ReLU / swish (-> "+ skip") chains of the reference's hot path:
//   stp3/layers/convolutions.py:183-280 (UpsamplingConcat, UpsamplingAdd, ASPP, DeepLabHead),
//   stp3/layers/temporal.py:252-273, 315-325, 426-489 (CausalConv3d, conv_1x1x1_norm_activated,
//   TemporalBlock), stp3/models/decoder.py:22-140 (ResNet-18 blocks, heads) and the MBConv blocks
//   of the EfficientNet trunk that stp3/models/encoder.py:57-97 drives; sync_batchnorm=True of
//   train.py:47 is served by the split statistics / apply interface.
//
// All of this is HBM-bound elementwise / reduction work: the kernels stream 16-byte channel
// vectors (8 bf16 / 4 f32) of the [rows][C] matrix, accumulate in fp32, reduce deterministically
// (two-stage partials, no atomics) and touch every tensor the minimum number of times:
//   forward : stats pass reads x once; apply pass reads x (+res) once and writes y once
//   backward: reduce pass reads dy, x (+res); apply pass reads dy, x (+res), writes dx (+dres)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <initializer_list>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

struct BnDims {
    int N, rows, C, ldx, ldy, ldr;
    int act, res_mode, has_sbias, has_oscale;
    int CL;     // channel lanes per row that belong to the tensors: C, or stp3_bn_dims::cpad (zero-padded rows)
};

constexpr int kThreads = 256;
constexpr int kUnroll = 4;   // rows per thread per iteration: independent 16-byte loads in flight

// ---- element access ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

template <typename T, int VEC> struct Io;
template <> struct Io<float, 4> {
    typedef float4 Raw;                              // what a load leaves in registers until the row is worked on
    static __device__ Raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ void unpack(const Raw& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    static __device__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ void store(float* p, const float* f) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    }
};
template <> struct Io<float, 1> {
    typedef float Raw;
    static __device__ Raw load_raw(const float* p) { return p[0]; }
    static __device__ void unpack(const Raw& v, float* f) { f[0] = v; }
    static __device__ void load(const float* p, float* f) { f[0] = p[0]; }
    static __device__ void store(float* p, const float* f) { p[0] = f[0]; }
};
template <> struct Io<uint16_t, 8> {
    typedef uint4 Raw;                               // 8 bf16 stay packed (4 registers, not 8) while the loads are in flight
    static __device__ Raw load_raw(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ void unpack(const Raw& v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
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
template <> struct Io<uint16_t, 1> {
    typedef uint16_t Raw;
    static __device__ Raw load_raw(const uint16_t* p) { return p[0]; }
    static __device__ void unpack(const Raw& v, float* f) { f[0] = bf2f(v); }
    static __device__ void load(const uint16_t* p, float* f) { f[0] = bf2f(p[0]); }
    static __device__ void store(uint16_t* p, const float* f) { p[0] = (uint16_t)pack_bf16(f[0], 0.f); }
};

// ---- thread -> (row lane, channel vector) mapping shared by all streaming kernels ---------------
// grid = (row blocks, N samples, channel tiles); a block covers CVB channel vectors x RL row lanes,
// RL a power of two so that the in-block reduction is a tree.
struct Map {
    int CV, CVB, RL, cv, rl;
    int nvalid;   // channels of this thread's vector that exist (< VEC only in the last vector of zero-padded rows)
    bool live;
};
// FULL (chosen for the largest maps, see plan()): every thread of the workgroup is a live row lane
// (RL = 256 / CVB, not rounded down to a power of two) -- with C = 144 that is 252 instead of 144 threads.
template <int VEC, bool FULL>
__device__ __forceinline__ Map make_map(const BnDims& d) {
    Map m;
    m.CV = (d.CL + VEC - 1) / VEC;
    m.CVB = min(m.CV, kThreads);
    int rl = 1;
    if (FULL) {
        rl = kThreads / m.CVB;
    } else {
        while (rl * 2 * m.CVB <= kThreads) rl *= 2;
    }
    m.RL = rl;
    const int cvb = threadIdx.x % m.CVB;
    m.rl = threadIdx.x / m.CVB;
    m.cv = blockIdx.z * m.CVB + cvb;
    m.live = m.rl < m.RL && m.cv < m.CV;
    m.nvalid = min(max(d.C - m.cv * VEC, 0), VEC);
    return m;
}
// Padding lanes (channels >= C of rows padded to stp3_bn_dims::cpad) are READ AS ZERO whatever the memory holds, so
// that what the kernels write there is exactly zero.  Compiled in (TAIL) only for launches with padded rows.
template <int VEC, bool TAIL>
__device__ __forceinline__ void mask_tail(const Map& m, float* f) {
    if (TAIL && m.nvalid < VEC) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (j >= m.nvalid) f[j] = 0.f;
    }
}

// The row loops are instantiated per (activation, residual mode) INSIDE each kernel: with the two as run-time values
// the element loop carried ~2 scalar branches per element, which made these streaming kernels instruction-bound.
template <int V> struct IntC { static constexpr int value = V; };
template <int ACT>
__device__ __forceinline__ float act_fwd_c(float v) {
    if (ACT == STP3_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == STP3_ACT_SWISH) return v * fast_sigmoid(v);
    return v;
}
template <int ACT>
__device__ __forceinline__ float act_grad_c(float pre) {
    if (ACT == STP3_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
    if (ACT == STP3_ACT_SWISH) {
        const float s = fast_sigmoid(pre);
        return s * (1.f + pre * (1.f - s));
    }
    return 1.f;
}
template <class F>
__device__ __forceinline__ void dispatch_res(int res_mode, F&& f) {
    if (res_mode == STP3_RES_NONE) f(IntC<STP3_RES_NONE>{});
    else if (res_mode == STP3_RES_BEFORE_ACT) f(IntC<STP3_RES_BEFORE_ACT>{});
    else f(IntC<STP3_RES_AFTER_ACT>{});
}
template <class F>
__device__ __forceinline__ void dispatch_act_res(int act, int res_mode, F&& f) {
    if (act == STP3_ACT_NONE) dispatch_res(res_mode, [&](auto r) { f(IntC<STP3_ACT_NONE>{}, r); });
    else if (act == STP3_ACT_RELU) dispatch_res(res_mode, [&](auto r) { f(IntC<STP3_ACT_RELU>{}, r); });
    else dispatch_res(res_mode, [&](auto r) { f(IntC<STP3_ACT_SWISH>{}, r); });
}

// In-block tree reduction over the row lanes of K values per thread, then one partial row per block:
// partial[((n * gridDim.x + bx) * K + k) * C + c]
// All K sums go through the tree TOGETHER (red holds K planes): one barrier per level instead of K -- the same additions in
// the same order as one tree per sum (bit-identical), a third of the barriers behind every bn_bwd_reduce launch.
template <int VEC, int K, bool FULL>
__device__ __forceinline__ void block_reduce_store(const BnDims& d, const Map& m, float (*acc)[VEC], float* red,
                                                   float* __restrict__ partial) {
    const int cvb = threadIdx.x % m.CVB;
    const int width = m.CVB * VEC;
    constexpr int plane = kThreads * VEC;
    __syncthreads();
    if (m.rl < m.RL) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[k * plane + m.rl * width + cvb * VEC + j] = m.live ? acc[k][j] : 0.f;
    }
    __syncthreads();
    int first = m.RL >> 1;
    if (FULL) {                                               // RL need not be a power of two
        int p2 = 1;
        while (p2 < m.RL) p2 <<= 1;
        first = p2 >> 1;
    }
    for (int s = first; s > 0; s >>= 1) {
        if (m.rl < s && m.rl + s < m.RL) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    red[k * plane + m.rl * width + cvb * VEC + j] += red[k * plane + (m.rl + s) * width + cvb * VEC + j];
        }
        __syncthreads();
    }
    if (m.rl == 0 && m.cv < m.CV) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float* out = partial + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * K + k) * d.C + m.cv * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (m.cv * VEC + j < d.C) out[j] = red[k * plane + cvb * VEC + j];
        }
    }
}

// ---- forward statistics: per-channel sum and sum of squares of (x + sbias) ------------------------
template <typename T, int VEC, bool FULL, bool TAIL>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(BnDims d, const T* __restrict__ x,
                                                            const float* __restrict__ sbias,
                                                            float* __restrict__ partial) {
    __shared__ float red[2 * kThreads * VEC];
    const Map m = make_map<VEC, FULL>(d);
    const int n = blockIdx.y;
    float acc[2][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = 0.f;
    if (m.live) {
        const int c0 = m.cv * VEC;
        float sb[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sb[j] = (d.has_sbias && c0 + j < d.C) ? sbias[(size_t)n * d.C + c0 + j] : 0.f;
        const T* xs = x + (size_t)n * d.rows * d.ldx + c0;
        const int step = gridDim.x * m.RL;
        for (int r = blockIdx.x * m.RL + m.rl; r < d.rows; r += kUnroll * step) {
            float v[kUnroll][VEC];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u)                       // kUnroll independent loads in flight
                if (r + u * step < d.rows) {
                    Io<T, VEC>::load(xs + (size_t)(r + u * step) * d.ldx, v[u]);
                    mask_tail<VEC, TAIL>(m, v[u]);
                }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (r + u * step < d.rows) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float t = v[u][j] + sb[j];
                        acc[0][j] += t;
                        acc[1][j] = fmaf(t, t, acc[1][j]);
                    }
                }
            }
        }
    }
    block_reduce_store<VEC, 2, FULL>(d, m, acc, red, partial);
}

// ---- sums over partial rows: out[g][i] = sum_p partial[(g * parts + p) * width + i], in double ------
// 8 columns x 32 part lanes per workgroup, four independent loads in flight per thread: these launches sit between
// every statistics / reduce pass and its apply pass (~300 per training step), so their latency is on the critical path
constexpr int kRedCols = 8, kRedLanes = kThreads / kRedCols;
__global__ __launch_bounds__(kThreads) void bn_reduce_partials_kernel(int parts, int width,
                                                                      const float* __restrict__ partial,
                                                                      float* __restrict__ out) {
    __shared__ double red[kThreads];
    const int il = threadIdx.x % kRedCols, pl = threadIdx.x / kRedCols;
    const int i = blockIdx.x * kRedCols + il;
    const int g = blockIdx.y;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < width) {
        const float* src = partial + (size_t)g * parts * width + i;
        int p = pl;
        for (; p + 3 * kRedLanes < parts; p += 4 * kRedLanes) {
            const float a = src[(size_t)p * width], b = src[(size_t)(p + kRedLanes) * width];
            const float c = src[(size_t)(p + 2 * kRedLanes) * width], d = src[(size_t)(p + 3 * kRedLanes) * width];
            s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
        }
        for (; p < parts; p += kRedLanes) s0 += (double)src[(size_t)p * width];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int st = kRedLanes / 2; st > 0; st >>= 1) {
        if (pl < st) red[threadIdx.x] += red[threadIdx.x + st * kRedCols];
        __syncthreads();
    }
    if (pl == 0 && i < width) out[(size_t)g * width + i] = (float)red[il];
}

// per-channel scale / shift from the (global) sums; block (0,0,*) row lane 0 also records mean / invstd
// and updates the running statistics (momentum, unbiased variance) exactly like nn.BatchNorm does
template <int VEC>
__device__ __forceinline__ void channel_affine(const BnDims& d, const Map& m, const float* __restrict__ sums,
                                               float inv_count, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps, float* mean, float* invstd,
                                               float* scale, float* shift) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = m.cv * VEC + j;
        if (c < d.C) {
            mean[j] = sums[c] * inv_count;
            const float var = fmaxf(sums[d.C + c] * inv_count - mean[j] * mean[j], 0.f);
            invstd[j] = 1.0f / sqrtf(var + eps);
            const float g = gamma ? gamma[c] : 1.f;
            scale[j] = g * invstd[j];
            shift[j] = (beta ? beta[c] : 0.f) - mean[j] * scale[j];
        } else {
            mean[j] = 0.f; invstd[j] = 0.f; scale[j] = 0.f; shift[j] = 0.f;
        }
    }
}

// ---- forward apply -------------------------------------------------------------------------------
// y = act(((x + sbias) - mean) * invstd * gamma + beta [+ res]) [* oscale[n]] [+ res]
// TRAIN: mean / invstd from `sums` (of `count` elements); else from the running statistics.
template <typename T, int VEC, bool TRAIN, bool FULL, bool TAIL>
__global__ __launch_bounds__(kThreads) void bn_apply_fwd_kernel(
    BnDims d, const T* __restrict__ x, const float* __restrict__ sbias, const T* __restrict__ res,
    const float* __restrict__ oscale, const float* __restrict__ sums, float inv_count, float unbias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, T* __restrict__ y) {
    const Map m = make_map<VEC, FULL>(d);
    if (!m.live) return;
    const int n = blockIdx.y;
    const int c0 = m.cv * VEC;
    float mean[VEC], invstd[VEC], scale[VEC], shift[VEC];
    if (TRAIN) {
        channel_affine<VEC>(d, m, sums, inv_count, gamma, beta, eps, mean, invstd, scale, shift);
        if (blockIdx.x == 0 && n == 0 && m.rl == 0) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = c0 + j;
                if (c < d.C) {
                    save_mean[c] = mean[j];
                    save_invstd[c] = invstd[j];
                    if (running_mean) {
                        const float var = fmaxf(sums[d.C + c] * inv_count - mean[j] * mean[j], 0.f);
                        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[j];
                        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * unbias;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = c0 + j;
            if (c < d.C) {
                const float is = 1.0f / sqrtf(running_var[c] + eps);
                scale[j] = (gamma ? gamma[c] : 1.f) * is;
                shift[j] = (beta ? beta[c] : 0.f) - running_mean[c] * scale[j];
            } else {
                scale[j] = shift[j] = 0.f;
            }
        }
    }
    if (d.has_sbias) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (c0 + j < d.C) shift[j] = fmaf(sbias[(size_t)n * d.C + c0 + j], scale[j], shift[j]);
    }
    const float os = d.has_oscale ? oscale[n] : 1.f;
    const T* xs = x + (size_t)n * d.rows * d.ldx + c0;
    const T* rs = d.res_mode ? res + (size_t)n * d.rows * d.ldr + c0 : nullptr;
    T* ys = y + (size_t)n * d.rows * d.ldy + c0;
    const int step = gridDim.x * m.RL;
    dispatch_act_res(d.act, d.res_mode, [&](auto act_c, auto res_c) {
        constexpr int ACT = decltype(act_c)::value, RESM = decltype(res_c)::value;
        typedef typename Io<T, VEC>::Raw Raw;
        for (int r = blockIdx.x * m.RL + m.rl; r < d.rows; r += kUnroll * step) {
            Raw xr[kUnroll], rr[kUnroll];                  // loads stay packed while they are in flight
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (r + u * step < d.rows) {
                    xr[u] = Io<T, VEC>::load_raw(xs + (size_t)(r + u * step) * d.ldx);
                    if (RESM != STP3_RES_NONE) rr[u] = Io<T, VEC>::load_raw(rs + (size_t)(r + u * step) * d.ldr);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (r + u * step < d.rows) {
                    float v[VEC], rv[VEC];
                    Io<T, VEC>::unpack(xr[u], v);
                    mask_tail<VEC, TAIL>(m, v);
                    if (RESM != STP3_RES_NONE) {
                        Io<T, VEC>::unpack(rr[u], rv);
                        mask_tail<VEC, TAIL>(m, rv);
                    }
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float t = fmaf(v[j], scale[j], shift[j]);
                        if (RESM == STP3_RES_BEFORE_ACT) t += rv[j];
                        t = act_fwd_c<ACT>(t) * os;
                        if (RESM == STP3_RES_AFTER_ACT) t += rv[j];
                        v[j] = t;
                    }
                    Io<T, VEC>::store(ys + (size_t)(r + u * step) * d.ldy, v);
                }
            }
        }
    });
}

// ---- backward reduce: per (sample, channel) sums of g, g * xhat and xhat -----------------------
// g = dy * oscale[n] * act'(pre), pre = xhat * gamma + beta [+ res]
template <typename T, int VEC, bool FULL, bool TAIL>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(
    BnDims d, const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ sbias,
    const T* __restrict__ res, const float* __restrict__ oscale, const float* __restrict__ mean_,
    const float* __restrict__ invstd_, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ partial) {
    __shared__ float red[3 * kThreads * VEC];
    const Map m = make_map<VEC, FULL>(d);
    const int n = blockIdx.y;
    float acc[3][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
    if (m.live) {
        const int c0 = m.cv * VEC;
        // xhat = ((x + sb) - mean) * invstd; the pre-activation as the apply pass (and the forward) form it: s x + t
        float mu[VEC], is[VEC], cs[VEC], ct[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = c0 + j;
            const bool ok = c < d.C;
            const float sb = (ok && d.has_sbias) ? sbias[(size_t)n * d.C + c] : 0.f;
            mu[j] = ok ? mean_[c] - sb : 0.f;
            is[j] = ok ? invstd_[c] : 0.f;
            const float ga = ok ? (gamma ? gamma[c] : 1.f) : 0.f;
            const float be = ok ? (beta ? beta[c] : 0.f) : 0.f;
            cs[j] = ga * is[j];
            ct[j] = be - mu[j] * cs[j];
        }
        const float os = d.has_oscale ? oscale[n] : 1.f;
        const T* xs = x + (size_t)n * d.rows * d.ldx + c0;
        const T* gs = dy + (size_t)n * d.rows * d.ldy + c0;
        const T* rs = d.res_mode == STP3_RES_BEFORE_ACT ? res + (size_t)n * d.rows * d.ldr + c0 : nullptr;
        const int step = gridDim.x * m.RL;
        constexpr int U = 3;                               // rows in flight per thread, loads kept packed until used
        typedef typename Io<T, VEC>::Raw Raw;
        // (the residual only matters when it is added in front of the activation)
        dispatch_act_res(d.act, rs ? STP3_RES_BEFORE_ACT : STP3_RES_NONE, [&](auto act_c, auto res_c) {
            constexpr int ACT = decltype(act_c)::value;
            constexpr bool PRE_RES = decltype(res_c)::value == STP3_RES_BEFORE_ACT;
            for (int r = blockIdx.x * m.RL + m.rl; r < d.rows; r += U * step) {
                Raw xr[U], gr[U], rr[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r + u * step < d.rows) {
                        xr[u] = Io<T, VEC>::load_raw(xs + (size_t)(r + u * step) * d.ldx);
                        gr[u] = Io<T, VEC>::load_raw(gs + (size_t)(r + u * step) * d.ldy);
                        if (PRE_RES) rr[u] = Io<T, VEC>::load_raw(rs + (size_t)(r + u * step) * d.ldr);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r + u * step < d.rows) {
                        float v[VEC], g[VEC], rv[VEC];
                        Io<T, VEC>::unpack(xr[u], v);
                        Io<T, VEC>::unpack(gr[u], g);
                        mask_tail<VEC, TAIL>(m, v);
                        mask_tail<VEC, TAIL>(m, g);
                        if (PRE_RES) {
                            Io<T, VEC>::unpack(rr[u], rv);
                            mask_tail<VEC, TAIL>(m, rv);
                        }
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            const float xh = (v[j] - mu[j]) * is[j];
                            float gg = g[j] * os;
                            if (ACT != STP3_ACT_NONE) {
                                float pre = fmaf(v[j], cs[j], ct[j]);
                                if (PRE_RES) pre += rv[j];
                                gg *= act_grad_c<ACT>(pre);
                            }
                            acc[0][j] += gg;
                            acc[1][j] = fmaf(gg, xh, acc[1][j]);
                            acc[2][j] += xh;                            // needed for the per-sample bias gradient
                        }
                    }
                }
            }
        });
    }
    block_reduce_store<VEC, 3, FULL>(d, m, acc, red, partial);
}

// ---- backward apply: dx = gamma * invstd * (g - sum(g)/M - xhat * sum(g*xhat)/M), dres ----------
// TRAIN == false (running statistics are constants): dx = gamma * invstd * g
template <typename T, int VEC, bool TRAIN, bool FULL, bool TAIL>
__global__ __launch_bounds__(kThreads) void bn_apply_bwd_kernel(
    BnDims d, const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ sbias,
    const T* __restrict__ res, const float* __restrict__ oscale, const float* __restrict__ mean_,
    const float* __restrict__ invstd_, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ gsums, float inv_count, T* __restrict__ dx, T* __restrict__ dres) {
    const Map m = make_map<VEC, FULL>(d);
    if (!m.live) return;
    const int n = blockIdx.y;
    const int c0 = m.cv * VEC;
    // FOUR constants per channel instead of (mean, invstd, gamma, beta, k0, k1): the pre-activation is the forward's
    // pre = s x + t (s = gamma invstd, t = beta - mean s) and dx = gamma invstd (g - k0 - xhat k1) = s g + a2 x + a3 with
    // a2 = -s invstd k1, a3 = -s k0 - a2 mean.  The kernel held 166 registers (3 waves per SIMD, 2.4 resident on average:
    // profiles/r03w_bn_pmc.txt) and a streaming kernel lives on the bytes it keeps in flight.
    float cs[VEC], ct[VEC], a2[VEC], a3[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = c0 + j;
        const bool ok = c < d.C;
        const float sb = (ok && d.has_sbias) ? sbias[(size_t)n * d.C + c] : 0.f;
        const float mu = ok ? mean_[c] - sb : 0.f;
        const float is = ok ? invstd_[c] : 0.f;
        const float ga = ok ? (gamma ? gamma[c] : 1.f) : 0.f;
        const float be = ok ? (beta ? beta[c] : 0.f) : 0.f;
        const float k0 = (TRAIN && ok) ? gsums[c] * inv_count : 0.f;
        const float k1 = (TRAIN && ok) ? gsums[d.C + c] * inv_count : 0.f;
        cs[j] = ga * is;
        ct[j] = be - mu * cs[j];
        a2[j] = -(cs[j] * is) * k1;
        a3[j] = -cs[j] * k0 - a2[j] * mu;
    }
    const float os = d.has_oscale ? oscale[n] : 1.f;
    const T* xs = x + (size_t)n * d.rows * d.ldx + c0;
    const T* gs = dy + (size_t)n * d.rows * d.ldy + c0;
    const bool pre_res = d.res_mode == STP3_RES_BEFORE_ACT;
    const T* rs = pre_res ? res + (size_t)n * d.rows * d.ldr + c0 : nullptr;
    T* dxs = dx + (size_t)n * d.rows * d.ldx + c0;
    T* drs = (pre_res && dres) ? dres + (size_t)n * d.rows * d.ldr + c0 : nullptr;
    const int step = gridDim.x * m.RL;
    constexpr int U = 4;                                   // rows in flight per thread: their loads stay PACKED until used
    typedef typename Io<T, VEC>::Raw Raw;
    dispatch_act_res(d.act, rs ? STP3_RES_BEFORE_ACT : STP3_RES_NONE, [&](auto act_c, auto res_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool PRE_RES = decltype(res_c)::value == STP3_RES_BEFORE_ACT;
        for (int r = blockIdx.x * m.RL + m.rl; r < d.rows; r += U * step) {
            Raw xr[U], gr[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * step < d.rows) {
                    xr[u] = Io<T, VEC>::load_raw(xs + (size_t)(r + u * step) * d.ldx);
                    gr[u] = Io<T, VEC>::load_raw(gs + (size_t)(r + u * step) * d.ldy);
                    if (PRE_RES) rr[u] = Io<T, VEC>::load_raw(rs + (size_t)(r + u * step) * d.ldr);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * step < d.rows) {
                    float v[VEC], g[VEC], rv[VEC];
                    Io<T, VEC>::unpack(xr[u], v);
                    Io<T, VEC>::unpack(gr[u], g);
                    mask_tail<VEC, TAIL>(m, v);
                    mask_tail<VEC, TAIL>(m, g);
                    if (PRE_RES) {
                        Io<T, VEC>::unpack(rr[u], rv);
                        mask_tail<VEC, TAIL>(m, rv);
                    }
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float gg = g[j] * os;
                        if (ACT != STP3_ACT_NONE) {
                            float pre = fmaf(v[j], cs[j], ct[j]);
                            if (PRE_RES) pre += rv[j];
                            gg *= act_grad_c<ACT>(pre);
                        }
                        g[j] = gg;                                     // gradient w.r.t. the pre-activation
                        v[j] = fmaf(cs[j], gg, fmaf(a2[j], v[j], a3[j]));
                    }
                    Io<T, VEC>::store(dxs + (size_t)(r + u * step) * d.ldx, v);
                    if (drs) Io<T, VEC>::store(drs + (size_t)(r + u * step) * d.ldr, g);
                }
            }
        }
    });
}

// ---- gradient of the per-sample bias from the backward sums (a handful of values: one launch instead of six torch
//      operators per layer): dsbias[n][c] = gamma invstd (S0[n] - rows k0 - S2[n] k1), k = gsums / count; evaluation mode
//      (running statistics are constants): gamma invstd S0[n] -------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bn_dsbias_kernel(int N, int C, float rows, const float* __restrict__ sample_sums,
                                                             const float* __restrict__ gsums, float inv_count,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ invstd, float* __restrict__ out) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float* s = sample_sums + (size_t)n * 3 * C;
    const float gi = (gamma ? gamma[c] : 1.f) * invstd[c];
    float v = s[c];
    if (gsums) {
        const float k0 = gsums[c] * inv_count, k1 = gsums[C + c] * inv_count;
        v = (v - rows * k0) - s[2 * C + c] * k1;
    }
    out[i] = gi * v;
}

// ---- y = x_0 + x_1 + ... + x_{n-1} (n <= 8 dense tensors of one layout): the gradients arriving at a tensor that feeds
//      several consumers, added in ONE pass with float32 accumulation (input order) and a single rounding -- instead of
//      n - 1 pairwise additions that each read two tensors, write one and round ----------------------------------------
struct SumPtrs {
    const void* p[8];
};
// plane (nullable): one more addend that is CONSTANT over the plane of every (sample, channel) -- the gradient of a whole-plane
// mean, [N][C] in the tensors' type -- for channels-last tensors [N][H*W][C]: vector i belongs to sample i / vps and to the
// channel vector i % cvs (vps = H*W*C / VEC vectors per sample, cvs = C / VEC).  Added last, in float32, before the one rounding.
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void sum_n_kernel(int n, size_t nvec, SumPtrs src, T* __restrict__ y,
                                                         const T* __restrict__ plane, unsigned vps, unsigned cvs) {
    typedef typename Io<T, VEC>::Raw Raw;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += 2 * stride) {
        const bool two = i + stride < nvec;
        Raw a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < n) {
                a[k] = Io<T, VEC>::load_raw((const T*)src.p[k] + i * VEC);
                if (two) b[k] = Io<T, VEC>::load_raw((const T*)src.p[k] + (i + stride) * VEC);
            }
        }
        float acc[VEC], acc2[VEC], v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = acc2[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < n) {
                Io<T, VEC>::unpack(a[k], v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += v[j];
                if (two) {
                    Io<T, VEC>::unpack(b[k], v);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc2[j] += v[j];
                }
            }
        }
        if (plane) {
            Io<T, VEC>::load(plane + ((size_t)(i / vps) * cvs + i % cvs) * VEC, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += v[j];
            if (two) {
                const size_t i2 = i + stride;
                Io<T, VEC>::load(plane + ((size_t)(i2 / vps) * cvs + i2 % cvs) * VEC, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc2[j] += v[j];
            }
        }
        Io<T, VEC>::store(y + i * VEC, acc);
        if (two) Io<T, VEC>::store(y + (i + stride) * VEC, acc2);
    }
}

// ---- host side --------------------------------------------------------------------------------
inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

struct Launch {
    BnDims d;
    int vec;       // 8 / 4 (vector path) or 1
    bool bf16;
    bool full;     // full-occupancy geometry: all row lanes live, ~4096 workgroups
    bool tail;     // zero-padded rows: the last channel vector masks its padding lanes
    dim3 grid;
    int parts;     // partial rows = N * grid.x
};

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// `per_cu`: workgroups of the kernel about to be launched that one CU keeps resident (its register allocation: bn_stats 64-72
// registers -> 7, bn_apply_fwd 78-92 -> 5, bn_bwd_reduce 120-122 -> 4, bn_apply_bwd 126-132 -> 3; pinned by
// tests/test_kernel_resources_cpu.py).  The grid is AT MOST one resident round of the chip's 256 CUs (four rounds in the
// full-occupancy geometry): these are streaming kernels whose blocks all take equally long, so a handful of blocks beyond a
// round runs alone behind it -- 12-sample maps got 86 * 12 = 1032 blocks for 1024 resident places (bn_bwd_reduce 80 -> 60 us at
// 12 x 128 x 200 x 200, the step 42.4 -> 41.4 ms with the rounding alone).
inline int plan(const stp3_bn_dims* p, Launch* L, std::initializer_list<const void*> vec_ptrs, int per_cu) {
    if (!p) return STP3_EINVAL;
    if (p->N <= 0 || p->rows <= 0 || p->C <= 0 || p->ldx < p->C || p->ldy < p->C) return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (p->act < STP3_ACT_NONE || p->act > STP3_ACT_SWISH) return STP3_EINVAL;
    if (p->res_mode < STP3_RES_NONE || p->res_mode > STP3_RES_AFTER_ACT) return STP3_EINVAL;
    if (p->res_mode != STP3_RES_NONE && p->ldr < p->C) return STP3_EINVAL;
    if ((int64_t)p->N * p->rows >= (1LL << 31)) return STP3_EUNSUP;
    // zero-padded rows: lanes [C, cpad) belong to x / y / dy / dx / res (/ dres) too
    const int CL = p->cpad ? p->cpad : p->C;
    if (CL < p->C || p->ldx < CL || p->ldy < CL || (p->res_mode != STP3_RES_NONE && p->ldr < CL)) return STP3_EINVAL;
    L->bf16 = p->dtype == STP3_DTYPE_BF16;
    const int wide = L->bf16 ? 8 : 4;
    bool vec_ok = CL % wide == 0 && p->ldx % wide == 0 && p->ldy % wide == 0 &&
                  (p->res_mode == STP3_RES_NONE || p->ldr % wide == 0);
    for (const void* q : vec_ptrs) vec_ok = vec_ok && (q == nullptr || aligned16(q));
    L->vec = vec_ok ? wide : 1;
    BnDims& d = L->d;
    d.N = p->N; d.rows = p->rows; d.C = p->C; d.ldx = p->ldx; d.ldy = p->ldy; d.ldr = p->ldr;
    d.act = p->act; d.res_mode = p->res_mode; d.has_sbias = p->has_sbias; d.has_oscale = p->has_oscale;
    d.CL = CL;
    // two workgroup geometries; the full-occupancy one pays off only on the largest maps (measured on the MI355X,
    // profiles/r02a_validate_switches.txt: 72 x 144 x 112 x 240 backward 1205 -> 771 us, smaller maps 0-40 % slower)
    L->tail = CL != p->C;
    L->full = !L->tail && (int64_t)p->N * p->rows * p->C >= (int64_t)200 * 1000 * 1000;
    const int CV = (CL + L->vec - 1) / L->vec;
    const int CVB = CV < kThreads ? CV : kThreads;
    int RL = 1;
    if (L->full) {
        RL = kThreads / CVB;
    } else {
        while (RL * 2 * CVB <= kThreads) RL *= 2;
    }
    const int ctiles = (CV + CVB - 1) / CVB;
    // at most one resident round (four in the full-occupancy geometry), at least 8 rows per row lane
    const int target = 256 * per_cu * (L->full ? 4 : 1);
    int bx = target / (p->N * ctiles);
    const int max_bx = (p->rows + RL * 8 - 1) / (RL * 8);
    if (bx > max_bx) bx = max_bx;
    if (bx < 1) bx = 1;
    if (bx > STP3_BN_MAX_ROW_BLOCKS) bx = STP3_BN_MAX_ROW_BLOCKS;
    L->grid = dim3(bx, p->N, ctiles);
    L->parts = p->N * bx;
    return STP3_OK;
}

// run CALL with `T` / `VEC` / `FULL` bound to the launch's element type, vector width and geometry
#define BN_SWITCH_V(L, F, TL, ...)                                              \
    {                                                                           \
        constexpr bool FULL = F;                                                \
        constexpr bool TAIL = TL;                                               \
        if ((L).bf16) {                                                         \
            if ((L).vec == 8) { using T = uint16_t; constexpr int VEC = 8; __VA_ARGS__; } \
            else              { using T = uint16_t; constexpr int VEC = 1; __VA_ARGS__; } \
        } else {                                                                \
            if ((L).vec == 4) { using T = float; constexpr int VEC = 4; __VA_ARGS__; }    \
            else              { using T = float; constexpr int VEC = 1; __VA_ARGS__; }    \
        }                                                                       \
    }
// zero-padded rows (tail) always take the default geometry: one extra instantiation per kernel, not two
#define BN_SWITCH(L, ...)                                                       \
    do {                                                                        \
        if ((L).tail) BN_SWITCH_V(L, false, true, __VA_ARGS__)                  \
        else if ((L).full) BN_SWITCH_V(L, true, false, __VA_ARGS__)             \
        else BN_SWITCH_V(L, false, false, __VA_ARGS__)                          \
    } while (0)

inline size_t ws_bytes(const stp3_bn_dims* p) {
    return (size_t)p->N * STP3_BN_MAX_ROW_BLOCKS * 3 * p->C * sizeof(float);
}

// partial [groups][parts][width] -> out [groups][width]
inline void reduce_partials(int groups, int parts, int width, const float* partial, float* out, hipStream_t s) {
    hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3((width + kRedCols - 1) / kRedCols, groups), dim3(kThreads), 0, s,
                       parts, width, partial, out);
}

}  // namespace

extern "C" {

int stp3_bn_workspace_bytes(const stp3_bn_dims* p, size_t* bytes) {
    if (!p || !bytes || p->N <= 0 || p->C <= 0) return STP3_EINVAL;
    *bytes = ws_bytes(p);
    return STP3_OK;
}

int stp3_bn_stats(const stp3_bn_dims* p, const void* x, const float* sbias, void* workspace, size_t workspace_bytes,
                  float* sums, void* stream) {
    Launch L;
    int rc = plan(p, &L, {x}, 7);
    if (rc) return rc;
    if (!x || !workspace || !sums || (p->has_sbias && !sbias)) return STP3_EINVAL;
    if (workspace_bytes < ws_bytes(p)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    BN_SWITCH(L, hipLaunchKernelGGL((bn_stats_kernel<T, VEC, FULL, TAIL>), L.grid, dim3(kThreads), 0, s, L.d, (const T*)x, sbias,
                                     partial));
    reduce_partials(1, L.parts, 2 * p->C, partial, sums, s);
    return status();
}

int stp3_bn_apply_fwd(const stp3_bn_dims* p, const void* x, const float* sbias, const void* res, const float* oscale,
                      const float* sums, double count, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                      void* y, void* stream) {
    Launch L;
    int rc = plan(p, &L, {x, res, y}, 5);
    if (rc) return rc;
    if (!x || !y || (p->has_sbias && !sbias) || (p->res_mode != STP3_RES_NONE && !res) ||
        (p->has_oscale && !oscale))
        return STP3_EINVAL;
    const bool train = sums != nullptr;
    if (train && (!save_mean || !save_invstd || !(count >= 1.0))) return STP3_EINVAL;
    if (train && ((running_mean == nullptr) != (running_var == nullptr))) return STP3_EINVAL;
    if (!train && (!running_mean || !running_var)) return STP3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const float inv_count = train ? (float)(1.0 / count) : 0.f;
    const float unbias = (train && count > 1.0) ? (float)(count / (count - 1.0)) : 1.f;
    if (train)
        BN_SWITCH(L, hipLaunchKernelGGL((bn_apply_fwd_kernel<T, VEC, true, FULL, TAIL>), L.grid, dim3(kThreads), 0, s, L.d,
                                         (const T*)x, sbias, (const T*)res, oscale, sums, inv_count, unbias, gamma,
                                         beta, eps, momentum, running_mean, running_var, save_mean, save_invstd,
                                         (T*)y));
    else
        BN_SWITCH(L, hipLaunchKernelGGL((bn_apply_fwd_kernel<T, VEC, false, FULL, TAIL>), L.grid, dim3(kThreads), 0, s, L.d,
                                         (const T*)x, sbias, (const T*)res, oscale, sums, inv_count, unbias, gamma,
                                         beta, eps, momentum, running_mean, running_var, save_mean, save_invstd,
                                         (T*)y));
    return status();
}

int stp3_bn_bwd_reduce(const stp3_bn_dims* p, const void* dy, const void* x, const float* sbias, const void* res,
                       const float* oscale, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, void* workspace, size_t workspace_bytes, float* sample_sums,
                       float* sums, void* stream) {
    Launch L;
    int rc = plan(p, &L, {dy, x, res}, 4);
    if (rc) return rc;
    if (!dy || !x || !mean || !invstd || !workspace || !sample_sums || !sums || (p->has_sbias && !sbias) ||
        (p->res_mode == STP3_RES_BEFORE_ACT && !res) || (p->has_oscale && !oscale))
        return STP3_EINVAL;
    if (workspace_bytes < ws_bytes(p)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    BN_SWITCH(L, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, FULL, TAIL>), L.grid, dim3(kThreads), 0, s, L.d, (const T*)dy,
                                     (const T*)x, sbias, (const T*)res, oscale, mean, invstd, gamma, beta, partial));
    if (p->has_sbias) {
        reduce_partials(p->N, (int)L.grid.x, 3 * p->C, partial, sample_sums, s);     // [N][3][C]
        reduce_partials(1, p->N, 3 * p->C, sample_sums, sums, s);                     // [3][C]
    } else {
        // the per-sample sums only serve the gradient of the per-sample bias: without one, a single launch
        reduce_partials(1, p->N * (int)L.grid.x, 3 * p->C, partial, sums, s);
    }
    return status();
}

int stp3_bn_apply_bwd(const stp3_bn_dims* p, const void* dy, const void* x, const float* sbias, const void* res,
                      const float* oscale, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, const float* sums, double count, void* dx, void* dres, void* stream) {
    Launch L;
    int rc = plan(p, &L, {dy, x, res, dx, dres}, 3);
    if (rc) return rc;
    if (!dy || !x || !mean || !invstd || !dx || (p->has_sbias && !sbias) ||
        (p->res_mode == STP3_RES_BEFORE_ACT && !res) || (p->has_oscale && !oscale))
        return STP3_EINVAL;
    const bool train = sums != nullptr;
    if (train && !(count >= 1.0)) return STP3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const float inv_count = train ? (float)(1.0 / count) : 0.f;
    if (train)
        BN_SWITCH(L, hipLaunchKernelGGL((bn_apply_bwd_kernel<T, VEC, true, FULL, TAIL>), L.grid, dim3(kThreads), 0, s, L.d,
                                         (const T*)dy, (const T*)x, sbias, (const T*)res, oscale, mean, invstd, gamma,
                                         beta, sums, inv_count, (T*)dx, (T*)dres));
    else
        BN_SWITCH(L, hipLaunchKernelGGL((bn_apply_bwd_kernel<T, VEC, false, FULL, TAIL>), L.grid, dim3(kThreads), 0, s, L.d,
                                         (const T*)dy, (const T*)x, sbias, (const T*)res, oscale, mean, invstd, gamma,
                                         beta, sums, inv_count, (T*)dx, (T*)dres));
    return status();
}


int stp3_bn_dsbias(int32_t N, int32_t C, int32_t rows, const float* sample_sums, const float* gsums, double count,
                   const float* gamma, const float* invstd, float* dsbias, void* stream) {
    if (N <= 0 || C <= 0 || rows <= 0 || !sample_sums || !invstd || !dsbias) return STP3_EINVAL;
    if (gsums && !(count >= 1.0)) return STP3_EINVAL;
    if ((int64_t)N * C >= (1LL << 31)) return STP3_EUNSUP;
    hipLaunchKernelGGL(bn_dsbias_kernel, dim3((unsigned)(((int64_t)N * C + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, N, C, (float)rows, sample_sums, gsums, gsums ? (float)(1.0 / count) : 0.f, gamma,
                       invstd, dsbias);
    return status();
}

static int sum_n_run(int32_t n, int64_t numel, int32_t dtype, const void* const* src, const void* plane, int64_t per_sample,
                     int32_t C, void* y, void* stream) {
    if (n < 1 || n > 8 || numel <= 0 || !src || !y) return STP3_EINVAL;
    if (dtype != STP3_DTYPE_F32 && dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (plane && (C <= 0 || per_sample <= 0 || per_sample % C || numel % per_sample || per_sample >= (1LL << 32))) return STP3_EINVAL;
    SumPtrs ptrs;
    const int wide = dtype == STP3_DTYPE_BF16 ? 8 : 4;
    bool vec = numel % wide == 0 && aligned16(y) && (!plane || (C % wide == 0 && aligned16(plane)));
    for (int k = 0; k < 8; ++k) {
        ptrs.p[k] = k < n ? src[k] : nullptr;
        if (k < n && !src[k]) return STP3_EINVAL;
        vec = vec && (k >= n || aligned16(src[k]));
    }
    const size_t nvec = vec ? (size_t)(numel / wide) : (size_t)numel;
    size_t blocks = (nvec + 2 * kThreads - 1) / (2 * kThreads);
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = (hipStream_t)stream;
    const int w = vec ? wide : 1;
    const unsigned vps = plane ? (unsigned)(per_sample / w) : 1u, cvs = plane ? (unsigned)(C / w) : 1u;
    if (dtype == STP3_DTYPE_BF16) {
        if (vec) hipLaunchKernelGGL((sum_n_kernel<uint16_t, 8>), dim3((unsigned)blocks), dim3(kThreads), 0, s, n, nvec, ptrs, (uint16_t*)y, (const uint16_t*)plane, vps, cvs);
        else hipLaunchKernelGGL((sum_n_kernel<uint16_t, 1>), dim3((unsigned)blocks), dim3(kThreads), 0, s, n, nvec, ptrs, (uint16_t*)y, (const uint16_t*)plane, vps, cvs);
    } else {
        if (vec) hipLaunchKernelGGL((sum_n_kernel<float, 4>), dim3((unsigned)blocks), dim3(kThreads), 0, s, n, nvec, ptrs, (float*)y, (const float*)plane, vps, cvs);
        else hipLaunchKernelGGL((sum_n_kernel<float, 1>), dim3((unsigned)blocks), dim3(kThreads), 0, s, n, nvec, ptrs, (float*)y, (const float*)plane, vps, cvs);
    }
    return status();
}

int stp3_sum_n(int32_t n, int64_t numel, int32_t dtype, const void* const* src, void* y, void* stream) {
    return sum_n_run(n, numel, dtype, src, nullptr, 0, 0, y, stream);
}

int stp3_sum_n_plane(int32_t n, int64_t numel, int32_t dtype, const void* const* src, const void* plane, int64_t per_sample,
                     int32_t C, void* y, void* stream) {
    if (!plane) return STP3_EINVAL;
    return sum_n_run(n, numel, dtype, src, plane, per_sample, C, y, stream);
}

// ---- single-process composites: the same launches as the split calls above, one host crossing each ----
int stp3_bn_fwd_train(const stp3_bn_dims* p, const void* x, const float* sbias, const void* res, const float* oscale,
                      const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* stat_buf, void* workspace, size_t workspace_bytes, void* y,
                      void* stream) {
    if (!p || !stat_buf) return STP3_EINVAL;
    const int C = p->C;
    int rc = stp3_bn_stats(p, x, sbias, workspace, workspace_bytes, stat_buf, stream);
    if (rc) return rc;
    return stp3_bn_apply_fwd(p, x, sbias, res, oscale, stat_buf, (double)p->N * p->rows, gamma, beta, eps, momentum,
                             running_mean, running_var, stat_buf + 2 * C, stat_buf + 3 * C, y, stream);
}

int stp3_bn_bwd_train(const stp3_bn_dims* p, const void* dy, const void* x, const float* sbias, const void* res,
                      const float* oscale, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, void* workspace, size_t workspace_bytes, float* sum_buf, void* dx, void* dres,
                      void* stream) {
    if (!p || !sum_buf) return STP3_EINVAL;
    float* sample_sums = sum_buf;                                  // [N][3][C]
    float* sums = sum_buf + (size_t)p->N * 3 * p->C;               // [3][C]
    int rc = stp3_bn_bwd_reduce(p, dy, x, sbias, res, oscale, mean, invstd, gamma, beta, workspace, workspace_bytes,
                                sample_sums, sums, stream);
    if (rc) return rc;
    return stp3_bn_apply_bwd(p, dy, x, sbias, res, oscale, mean, invstd, gamma, beta, sums, (double)p->N * p->rows, dx,
                             dres, stream);
}

}  // extern "C"
