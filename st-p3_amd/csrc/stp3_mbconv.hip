// stp3_mbconv.hip -- the passes over the EXPANDED tensor of an EfficientNet MBConv block that remain once BatchNorm,
// swish and the squeeze-excite gate are applied where their consumer loads its operand (gfx950, channels-last).
//
// Reference chain (efficientnet_pytorch MBConvBlock as driven by stp3/models/encoder.py:57-97):
//     E0 = expand(x) -> E1 = swish(BN0(E0)) -> E2 = depthwise(E1) -> S = swish(BN1(E2)) -> g = sigmoid(MLP(mean S))
//     -> A = S * g -> y = BN2(project(A)) (+ x)
// Executed literally that is 12 passes over the expanded tensor forward and 22 backward; with
// the kernels of this file, the statistics epilogue of the depthwise kernel (stp3_dwconv.hip) and the operand
// transforms of the pointwise kernels (stp3_conv.hip: stp3_conv2d_fwd_pre / stp3_conv2d_wgrad_pre) S and A are never
// written: forward 7 passes, backward 18.
//
//   stp3_bn_finalize       sums -> per-channel (scale, shift, mean, invstd) + running statistics: what a consumer that
//                          applies the BatchNorm on load needs, computed once instead of per thread
//   stp3_se_pool_act       squeeze of S without S:  out[n][c] = sum_r act(scale[c] * x[n][r][c] + shift[c])
//   stp3_mbconv_bwd_reduce ONE pass over (dA, E2) for everything the backward needs per (sample, channel):
//                          gate gradient sum dA*S, and the four sums from which the BatchNorm-1 reductions follow for
//                          ANY per-sample gate / pooled gradient (the SE backward needs the first before it can
//                          deliver the latter -- two dependent passes in the literal chain)
//   stp3_mbconv_bwd_coef   those sums + gate + dpooled -> sum g, sum g*xhat  ([2][C]: dbeta1, dgamma1)
//   stp3_mbconv_bwd_apply  dE2 = gamma1*invstd1 * (g - k0 - xhat*k1),  g = (dA*gate + dpooled) * act'(pre)
// HBM-bound streaming kernels: 16-byte channel vectors, float32 arithmetic, deterministic two-stage reductions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <initializer_list>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

constexpr int kT = 256;
constexpr int kMaxBx = 64;          // row blocks per sample (partial rows of the two-stage reductions)

struct MbDims {
    int N, rows, C, ldx, ldg;       // ldx: row stride of x (E2), ldg: of dA / dx
};

template <typename T, int VEC> struct Io3;
template <> struct Io3<float, 4> {
    typedef float4 Raw;                              // a load as it sits in registers until its row is worked on
    static __device__ Raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ void unpack(const Raw& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    static __device__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ void store(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct Io3<uint16_t, 8> {
    typedef uint4 Raw;                               // 8 bf16 packed: 4 registers instead of 8 while in flight
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

template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
    if (ACT == STP3_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == STP3_ACT_SWISH) return v * fast_sigmoid(v);
    return v;
}
// value and derivative of the activation at `pre` in one go (one sigmoid)
template <int ACT>
__device__ __forceinline__ void act_both(float pre, float& val, float& der) {
    if (ACT == STP3_ACT_SWISH) {
        const float s = fast_sigmoid(pre);
        val = pre * s;
        der = s * (1.f + pre * (1.f - s));
    } else if (ACT == STP3_ACT_RELU) {
        val = fmaxf(pre, 0.f);
        der = pre > 0.f ? 1.f : 0.f;
    } else {
        val = pre;
        der = 1.f;
    }
}

// thread -> (row lane, channel vector); grid = (row blocks, N, channel tiles); RL a power of two
template <int VEC>
__device__ __forceinline__ void mb_map(int C, int& cv, int& rl, int& RL, int& CVB, bool& live) {
    const int CV = C / VEC;
    CVB = min(CV, kT);
    RL = 1;
    while (RL * 2 * CVB <= kT) RL *= 2;
    const int cvb = threadIdx.x % CVB;
    rl = threadIdx.x / CVB;
    cv = blockIdx.z * CVB + cvb;
    live = rl < RL && cv < CV;
}

// tree reduction of K accumulator vectors over the row lanes; row lane 0 writes partial[((n * gridDim.x + bx) * K + k) * C + c]
template <int VEC, int K>
__device__ __forceinline__ void mb_block_reduce(int C, int cv, int rl, int RL, int CVB, bool live, float (*acc)[VEC], float* red,
                                                float* __restrict__ partial) {
    const int cvb = threadIdx.x % CVB;
    const int width = CVB * VEC;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        __syncthreads();
        if (rl < RL) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[rl * width + cvb * VEC + j] = live ? acc[k][j] : 0.f;
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
            float* out = partial + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * K + k) * C + cv * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j) out[j] = red[cvb * VEC + j];
        }
    }
}

// sum_p partial[(g * parts + p) * width + i], double accumulation, fixed order: 64 columns x 4 part lanes; the result goes
// to out[g][i] (K == 1) or, for width = K * C, to out[k][g][c] with i = k * C + c (quantity-major: each of the K
// quantities is a contiguous [groups][C] matrix)
__global__ __launch_bounds__(kT) void mb_reduce_kernel(int parts, int width, int K, const float* __restrict__ partial,
                                                       float* __restrict__ out) {
    __shared__ double red[kT];
    const int il = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    const int g = blockIdx.y;
    double s0 = 0.0, s1 = 0.0;
    if (i < width) {
        const float* src = partial + (size_t)g * parts * width + i;
        int p = pl;
        for (; p + 4 < parts; p += 8) {
            const float a = src[(size_t)p * width], b = src[(size_t)(p + 4) * width];
            s0 += (double)a; s1 += (double)b;
        }
        for (; p < parts; p += 4) s0 += (double)src[(size_t)p * width];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (pl == 0 && i < width) {
        const int C = width / K;
        const int k = i / C, c = i - k * C;
        out[((size_t)k * gridDim.y + g) * C + c] = (float)((red[il] + red[64 + il]) + (red[128 + il] + red[192 + il]));
    }
}

// ---- stp3_bn_finalize ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void bn_finalize_kernel(int C, const float* __restrict__ sums, float inv_count, float unbias,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ coef) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= C) return;
    // the arithmetic of stp3_bnact.hip's channel_affine, so that a layer gives the same bits fused or not
    const float mean = sums[c] * inv_count;
    const float var = fmaxf(sums[C + c] * inv_count - mean * mean, 0.f);
    const float invstd = 1.0f / sqrtf(var + eps);
    const float scale = (gamma ? gamma[c] : 1.f) * invstd;
    coef[c] = scale;
    coef[C + c] = (beta ? beta[c] : 0.f) - mean * scale;
    coef[2 * C + c] = mean;
    coef[3 * C + c] = invstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * unbias;
    }
}

// ---- stp3_se_pool_act ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(kT) void se_pool_act_kernel(MbDims d, const T* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ partial) {
    __shared__ float red[kT * VEC];
    int cv, rl, RL, CVB;
    bool live;
    mb_map<VEC>(d.C, cv, rl, RL, CVB, live);
    const int n = blockIdx.y;
    float acc[1][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[0][j] = 0.f;
    if (live) {
        float sc[VEC], sh[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { sc[j] = scale[cv * VEC + j]; sh[j] = shift[cv * VEC + j]; }
        const T* xs = x + (size_t)n * d.rows * d.ldx + cv * VEC;
        const int step = gridDim.x * RL;
        for (int r = blockIdx.x * RL + rl; r < d.rows; r += 4 * step) {
            float a[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + u * step < d.rows) Io3<T, VEC>::load(xs + (size_t)(r + u * step) * d.ldx, a[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * step < d.rows) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[0][j] += act_fwd<ACT>(fmaf(a[u][j], sc[j], sh[j]));
                }
            }
        }
    }
    mb_block_reduce<VEC, 1>(d.C, cv, rl, RL, CVB, live, acc, red, partial);
}

// ---- stp3_mbconv_bwd_reduce ---------------------------------------------------------------------------------------
// sums5 [5][N][C], per (n, c):  [0] sum dA*S   [1] sum dA*S'   [2] sum dA*S'*xhat   [3] sum S'   [4] sum S'*xhat
// S = act(pre), S' = act'(pre), pre = scale*x + shift, xhat = (x - mean) * invstd
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(kT) void mbconv_bwd_reduce_kernel(MbDims d, const T* __restrict__ da, const T* __restrict__ x,
                                                               const float* __restrict__ coef, float* __restrict__ partial) {
    __shared__ float red[kT * VEC];
    int cv, rl, RL, CVB;
    bool live;
    mb_map<VEC>(d.C, cv, rl, RL, CVB, live);
    const int n = blockIdx.y;
    float acc[5][VEC];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
    if (live) {
        float sc[VEC], sh[VEC], mu[VEC], is[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = cv * VEC + j;
            sc[j] = coef[c]; sh[j] = coef[d.C + c]; mu[j] = coef[2 * d.C + c]; is[j] = coef[3 * d.C + c];
        }
        const T* xs = x + (size_t)n * d.rows * d.ldx + cv * VEC;
        const T* gs = da + (size_t)n * d.rows * d.ldg + cv * VEC;
        const int step = gridDim.x * RL;
        for (int r = blockIdx.x * RL + rl; r < d.rows; r += 2 * step) {
            float v[2][VEC], g[2][VEC];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r + u * step < d.rows) {
                    Io3<T, VEC>::load(xs + (size_t)(r + u * step) * d.ldx, v[u]);
                    Io3<T, VEC>::load(gs + (size_t)(r + u * step) * d.ldg, g[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r + u * step < d.rows) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float val, der;
                        act_both<ACT>(fmaf(v[u][j], sc[j], sh[j]), val, der);
                        const float xh = (v[u][j] - mu[j]) * is[j];
                        const float gd = g[u][j] * der;
                        acc[0][j] = fmaf(g[u][j], val, acc[0][j]);
                        acc[1][j] += gd;
                        acc[2][j] = fmaf(gd, xh, acc[2][j]);
                        acc[3][j] += der;
                        acc[4][j] = fmaf(der, xh, acc[4][j]);
                    }
                }
            }
        }
    }
    mb_block_reduce<VEC, 5>(d.C, cv, rl, RL, CVB, live, acc, red, partial);
}

// ---- stp3_mbconv_bwd_coef: gsums[0][c] = sum_n gate*P1 + dpooled*P3 ; gsums[1][c] = sum_n gate*P2 + dpooled*P4 -----
// 16 channels x 16 sample lanes per workgroup (one thread per channel walking the 72 samples was a 26 us chain of
// dependent strided loads); the sample lanes are summed in a fixed tree: deterministic
constexpr int kCoefCh = 16, kCoefLanes = kT / kCoefCh;
__global__ __launch_bounds__(kT) void mbconv_bwd_coef_kernel(int N, int C, const float* __restrict__ sums5,
                                                             const float* __restrict__ gate, const float* __restrict__ dpooled,
                                                             float* __restrict__ gsums) {
    __shared__ double red[2][kT];
    const int cl = threadIdx.x % kCoefCh, nl = threadIdx.x / kCoefCh;
    const int c = blockIdx.x * kCoefCh + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        const size_t q = (size_t)N * C;
        for (int n = nl; n < N; n += kCoefLanes) {
            const float* p = sums5 + (size_t)n * C + c;             // [5][N][C]
            const float g = gate ? gate[(size_t)n * C + c] : 1.f;
            const float dp = dpooled ? dpooled[(size_t)n * C + c] : 0.f;
            s0 += (double)(g * p[q] + dp * p[3 * q]);
            s1 += (double)(g * p[2 * q] + dp * p[4 * q]);
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int st = kCoefLanes / 2; st > 0; st >>= 1) {
        if (nl < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st * kCoefCh];
            red[1][threadIdx.x] += red[1][threadIdx.x + st * kCoefCh];
        }
        __syncthreads();
    }
    if (nl == 0 && c < C) {
        gsums[c] = (float)red[0][cl];
        gsums[C + c] = (float)red[1][cl];
    }
}

// ---- stp3_mbconv_bwd_apply ----------------------------------------------------------------------------------------
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(kT) void mbconv_bwd_apply_kernel(MbDims d, const T* __restrict__ da, const T* __restrict__ x,
                                                              const float* __restrict__ coef, const float* __restrict__ gate,
                                                              const float* __restrict__ dpooled,
                                                              const float* __restrict__ gsums, float inv_count,
                                                              T* __restrict__ dx) {
    int cv, rl, RL, CVB;
    bool live;
    mb_map<VEC>(d.C, cv, rl, RL, CVB, live);
    if (!live) return;
    const int n = blockIdx.y;
    // dx = scale * (gg - k0 - xhat * k1) with xhat = (x - mean) * invstd  ==  scale * gg - e1 * x - e0:
    // six per-channel constants instead of eight (176 -> ~150 registers: a third wave per SIMD)
    float sc[VEC], sh[VEC], ga[VEC], dp[VEC], e0[VEC], e1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = cv * VEC + j;
        sc[j] = coef[c]; sh[j] = coef[d.C + c];
        const float mu = coef[2 * d.C + c], is = coef[3 * d.C + c];
        ga[j] = gate ? gate[(size_t)n * d.C + c] : 1.f;
        dp[j] = dpooled ? dpooled[(size_t)n * d.C + c] : 0.f;
        const float k0 = gsums[c] * inv_count, k1 = gsums[d.C + c] * inv_count;
        e1[j] = sc[j] * k1 * is;
        e0[j] = sc[j] * (k0 - k1 * mu * is);
    }
    const T* xs = x + (size_t)n * d.rows * d.ldx + cv * VEC;
    const T* gs = da + (size_t)n * d.rows * d.ldg + cv * VEC;
    T* os = dx + (size_t)n * d.rows * d.ldg + cv * VEC;
    const int step = gridDim.x * RL;
    // four rows in flight per thread, their loads PACKED until the row is worked on (see csrc/stp3_bnact.hip: a streaming
    // kernel lives on the bytes it keeps in flight; unpacked, two rows at 144 registers left 3 waves per SIMD)
    constexpr int U = 4;
    typedef typename Io3<T, VEC>::Raw Raw;
    for (int r = blockIdx.x * RL + rl; r < d.rows; r += U * step) {
        Raw xr[U], gr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * step < d.rows) {
                xr[u] = Io3<T, VEC>::load_raw(xs + (size_t)(r + u * step) * d.ldx);
                gr[u] = Io3<T, VEC>::load_raw(gs + (size_t)(r + u * step) * d.ldg);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * step < d.rows) {
                float v[VEC], g[VEC];
                Io3<T, VEC>::unpack(xr[u], v);
                Io3<T, VEC>::unpack(gr[u], g);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float val, der;
                    act_both<ACT>(fmaf(v[j], sc[j], sh[j]), val, der);
                    const float gg = fmaf(g[j], ga[j], dp[j]) * der;
                    g[j] = fmaf(sc[j], gg, -fmaf(e1[j], v[j], e0[j]));
                }
                Io3<T, VEC>::store(os + (size_t)(r + u * step) * d.ldg, g);
            }
        }
    }
}

// ---- stp3_mbconv_scale_act:  y = act(scale[c] * x + shift[c]) * gate[n][c]  (BatchNorm-1 + swish + SE gate, one pass) ----
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(kT) void mbconv_scale_act_kernel(MbDims d, const T* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ gate,
                                                              T* __restrict__ y) {
    int cv, rl, RL, CVB;
    bool live;
    mb_map<VEC>(d.C, cv, rl, RL, CVB, live);
    if (!live) return;
    const int n = blockIdx.y;
    float sc[VEC], sh[VEC], ga[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = cv * VEC + j;
        sc[j] = scale[c]; sh[j] = shift[c];
        ga[j] = gate ? gate[(size_t)n * d.C + c] : 1.f;
    }
    const T* xs = x + (size_t)n * d.rows * d.ldx + cv * VEC;
    T* ys = y + (size_t)n * d.rows * d.ldg + cv * VEC;
    const int step = gridDim.x * RL;
    constexpr int U = 6;                                   // rows in flight per thread, packed until used
    typedef typename Io3<T, VEC>::Raw Raw;
    for (int r = blockIdx.x * RL + rl; r < d.rows; r += U * step) {
        Raw xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u * step < d.rows) xr[u] = Io3<T, VEC>::load_raw(xs + (size_t)(r + u * step) * d.ldx);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * step < d.rows) {
                float v[VEC];
                Io3<T, VEC>::unpack(xr[u], v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[j] = act_fwd<ACT>(fmaf(v[j], sc[j], sh[j])) * ga[j];
                Io3<T, VEC>::store(ys + (size_t)(r + u * step) * d.ldg, v);
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct MbPlan {
    MbDims d;
    bool bf16;
    dim3 grid;
};

// `per_cu`: resident workgroups per CU of the kernel about to be launched (se_pool_act / mbconv_scale_act 64-72 registers
// -> 7, mbconv_bwd_reduce 120-128 -> 4, mbconv_bwd_apply 96-112 -> 4); the grid is at most `rounds` resident rounds of the
// chip (see plan() in stp3_bnact.hip)
inline int mb_plan(const stp3_se_dims* p, int ldg, MbPlan* P, std::initializer_list<const void*> ptrs, int per_cu, int rounds) {
    if (!p || p->N <= 0 || p->rows <= 0 || p->C <= 0 || p->ld < p->C || ldg < p->C) return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    P->bf16 = p->dtype == STP3_DTYPE_BF16;
    const int vec = P->bf16 ? 8 : 4;
    if (p->C % vec || p->ld % vec || ldg % vec) return STP3_EUNSUP;             // whole 16-byte channel vectors only
    for (const void* q : ptrs)
        if (q && ((uintptr_t)q & 15)) return STP3_EUNSUP;
    P->d.N = p->N; P->d.rows = p->rows; P->d.C = p->C; P->d.ldx = p->ld; P->d.ldg = ldg;
    const int CV = p->C / vec;
    const int CVB = CV < kT ? CV : kT;
    int RL = 1;
    while (RL * 2 * CVB <= kT) RL *= 2;
    const int ctiles = (CV + CVB - 1) / CVB;
    int bx = (256 * per_cu * rounds) / (p->N * ctiles);
    const int max_bx = (p->rows + RL * 8 - 1) / (RL * 8);
    if (bx > max_bx) bx = max_bx;
    if (bx < 1) bx = 1;
    if (bx > kMaxBx) bx = kMaxBx;
    P->grid = dim3(bx, p->N, ctiles);
    return STP3_OK;
}

#define MB_SWITCH(P, ACTV, ...)                                                                      \
    do {                                                                                             \
        if ((ACTV) == STP3_ACT_SWISH) {                                                              \
            constexpr int ACT = STP3_ACT_SWISH;                                                      \
            if ((P).bf16) { using T = uint16_t; constexpr int VEC = 8; __VA_ARGS__; }                \
            else          { using T = float; constexpr int VEC = 4; __VA_ARGS__; }                   \
        } else if ((ACTV) == STP3_ACT_RELU) {                                                        \
            constexpr int ACT = STP3_ACT_RELU;                                                       \
            if ((P).bf16) { using T = uint16_t; constexpr int VEC = 8; __VA_ARGS__; }                \
            else          { using T = float; constexpr int VEC = 4; __VA_ARGS__; }                   \
        } else {                                                                                     \
            constexpr int ACT = STP3_ACT_NONE;                                                       \
            if ((P).bf16) { using T = uint16_t; constexpr int VEC = 8; __VA_ARGS__; }                \
            else          { using T = float; constexpr int VEC = 4; __VA_ARGS__; }                   \
        }                                                                                            \
    } while (0)

inline int mb_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline bool act_ok(int act) { return act >= STP3_ACT_NONE && act <= STP3_ACT_SWISH; }

}  // namespace

extern "C" {

int stp3_bn_finalize(const float* sums, int32_t C, double count, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* coef, void* stream) {
    if (!sums || !coef || C <= 0 || !(count >= 1.0)) return STP3_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return STP3_EINVAL;
    const float inv_count = (float)(1.0 / count);
    const float unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + kT - 1) / kT), dim3(kT), 0, (hipStream_t)stream, (int)C, sums, inv_count,
                       unbias, gamma, beta, eps, momentum, running_mean, running_var, coef);
    return mb_status();
}

int stp3_mbconv_workspace_bytes(const stp3_se_dims* p, size_t* bytes) {
    if (!p || !bytes || p->N <= 0 || p->C <= 0) return STP3_EINVAL;
    *bytes = (size_t)p->N * kMaxBx * 5 * p->C * sizeof(float);
    return STP3_OK;
}

int stp3_se_pool_act(const stp3_se_dims* p, const void* x, const float* scale, const float* shift, int32_t act,
                     void* workspace, size_t workspace_bytes, float* out, void* stream) {
    MbPlan P;
    int rc = mb_plan(p, p ? p->ld : 0, &P, {x}, 7, 1);
    if (rc) return rc;
    if (!x || !scale || !shift || !workspace || !out || !act_ok(act)) return STP3_EINVAL;
    if (workspace_bytes < (size_t)p->N * kMaxBx * p->C * sizeof(float)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    MB_SWITCH(P, act, hipLaunchKernelGGL((se_pool_act_kernel<T, VEC, ACT>), P.grid, dim3(kT), 0, s, P.d, (const T*)x, scale, shift,
                                         partial));
    hipLaunchKernelGGL(mb_reduce_kernel, dim3((p->C + 63) / 64, p->N), dim3(kT), 0, s, (int)P.grid.x, p->C, 1, partial, out);
    return mb_status();
}

int stp3_mbconv_scale_act(const stp3_se_dims* p, int32_t ldy, const void* x, const float* scale, const float* shift, int32_t act,
                          const float* gate, void* y, void* stream) {
    MbPlan P;
    int rc = mb_plan(p, ldy, &P, {x, y}, 7, 1);
    if (rc) return rc;
    if (!x || !scale || !shift || !y || !act_ok(act)) return STP3_EINVAL;
    MB_SWITCH(P, act, hipLaunchKernelGGL((mbconv_scale_act_kernel<T, VEC, ACT>), P.grid, dim3(kT), 0, (hipStream_t)stream, P.d,
                                         (const T*)x, scale, shift, gate, (T*)y));
    return mb_status();
}

int stp3_mbconv_bwd_reduce(const stp3_se_dims* p, int32_t ldg, const void* da, const void* x, const float* coef, int32_t act,
                           void* workspace, size_t workspace_bytes, float* sums5, void* stream) {
    MbPlan P;
    int rc = mb_plan(p, ldg, &P, {da, x}, 4, 2);
    if (rc) return rc;
    if (!da || !x || !coef || !workspace || !sums5 || !act_ok(act)) return STP3_EINVAL;
    if (workspace_bytes < (size_t)p->N * kMaxBx * 5 * p->C * sizeof(float)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    MB_SWITCH(P, act, hipLaunchKernelGGL((mbconv_bwd_reduce_kernel<T, VEC, ACT>), P.grid, dim3(kT), 0, s, P.d, (const T*)da,
                                         (const T*)x, coef, partial));
    // partial [N][bx][5][C] -> sums5 [5][N][C]
    hipLaunchKernelGGL(mb_reduce_kernel, dim3((5 * p->C + 63) / 64, p->N), dim3(kT), 0, s, (int)P.grid.x, 5 * p->C, 5, partial,
                       sums5);
    return mb_status();
}

int stp3_mbconv_bwd_coef(int32_t N, int32_t C, const float* sums5, const float* gate, const float* dpooled, float* gsums,
                         void* stream) {
    if (N <= 0 || C <= 0 || !sums5 || !gsums) return STP3_EINVAL;
    hipLaunchKernelGGL(mbconv_bwd_coef_kernel, dim3((C + kCoefCh - 1) / kCoefCh), dim3(kT), 0, (hipStream_t)stream, (int)N, (int)C, sums5,
                       gate, dpooled, gsums);
    return mb_status();
}

int stp3_mbconv_bwd_apply(const stp3_se_dims* p, int32_t ldg, const void* da, const void* x, const float* coef, int32_t act,
                          const float* gate, const float* dpooled, const float* gsums, double count, void* dx, void* stream) {
    MbPlan P;
    int rc = mb_plan(p, ldg, &P, {da, x, dx}, 4, 2);
    if (rc) return rc;
    if (!da || !x || !coef || !gsums || !dx || !act_ok(act) || !(count >= 1.0)) return STP3_EINVAL;
    const float inv_count = (float)(1.0 / count);
    MB_SWITCH(P, act, hipLaunchKernelGGL((mbconv_bwd_apply_kernel<T, VEC, ACT>), P.grid, dim3(kT), 0, (hipStream_t)stream, P.d,
                                         (const T*)da, (const T*)x, coef, gate, dpooled, gsums, inv_count, (T*)dx));
    return mb_status();
}

}  // extern "C"
