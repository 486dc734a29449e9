// stp3_gru.hip -- the element-wise half of the convolutional GRU cells of the prediction stage on gfx950.
//
// Reference: stp3/layers/temporal.py:42-56 (SpatialGRU.gru_cell) and :118-145 (Dual_GRU.gru_cell_1 / _2), 24 cells per
// training step of Prediction.yml:
//     update, reset = sigmoid(conv_update([x, state]) + b0), sigmoid(conv_reset([x, state]) + b0)
//     tilde         = conv_state_tilde([x, (1 - reset) * state])
//     out           = (1 - update) * state + update * tilde
// The three convolutions run on stp3_conv.hip (update and reset as ONE convolution with 2C output channels).  What is left
// is a dozen element-wise torch launches per cell forward and two dozen backward -- casts, sigmoid, rsub, mul, cat, and the
// gradient additions of a tensor used three times (1.4 ms per cell in all, 35 ms per step).  Here: two launches forward
// (reset gate applied while the second convolution's operand [x, (1 - r) state] is assembled; the output blend) and two
// backward, float32 arithmetic on bf16 / float32 rows, one rounding per result.  HBM-bound streaming kernels.
//
// Row layouts (channels-last pixels):  xs, xs2, dxs2, acc [rows][Cx + C] = [x | state];  gates, dgates [rows][2C] =
// [update | reset] pre-activations;  tilde, dtilde, out [rows][C];  dout [rows][ld_dout >= C].
#include <hip/hip_runtime.h>
#include <initializer_list>
#include <stdint.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

struct GruDims {
    int64_t rows;
    int Cx, C;
    float b0;
};

constexpr int kThreads = 256;

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ void store(float* p, const float* f) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
    static __device__ void copy(float* dst, const float* src) { *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src); }
};
template <> struct Vec<uint16_t> {
    static constexpr int N = 8;
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
    static __device__ void copy(uint16_t* dst, const uint16_t* src) { *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src); }
};

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// xs2 = [x | (1 - sigmoid(reset + b0)) * state]: one thread per 16-byte channel vector of the [Cx + C] row
template <typename T>
__global__ __launch_bounds__(kThreads) void gru_reset_cat_fwd_kernel(GruDims d, const T* __restrict__ xs,
                                                                     const T* __restrict__ gates, T* __restrict__ xs2) {
    constexpr int VN = Vec<T>::N;
    const int W = d.Cx + d.C, WV = W / VN;
    const int64_t total = d.rows * WV;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / WV;
        const int c = (int)(i % WV) * VN;
        if (c < d.Cx) {
            Vec<T>::copy(xs2 + r * W + c, xs + r * W + c);
            continue;
        }
        float st[VN], rp[VN];
        Vec<T>::load(xs + r * W + c, st);
        Vec<T>::load(gates + r * 2 * d.C + d.C + (c - d.Cx), rp);
#pragma unroll
        for (int j = 0; j < VN; ++j) st[j] = (1.0f - sigmoidf(rp[j] + d.b0)) * st[j];
        Vec<T>::store(xs2 + r * W + c, st);
    }
}

// out = (1 - u) * state + u * tilde,  u = sigmoid(update + b0)
template <typename T>
__global__ __launch_bounds__(kThreads) void gru_output_fwd_kernel(GruDims d, const T* __restrict__ gates,
                                                                  const T* __restrict__ xs, const T* __restrict__ tilde,
                                                                  T* __restrict__ out) {
    constexpr int VN = Vec<T>::N;
    const int W = d.Cx + d.C, CV = d.C / VN;
    const int64_t total = d.rows * CV;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / CV;
        const int c = (int)(i % CV) * VN;
        float up[VN], st[VN], tl[VN];
        Vec<T>::load(gates + r * 2 * d.C + c, up);
        Vec<T>::load(xs + r * W + d.Cx + c, st);
        Vec<T>::load(tilde + r * d.C + c, tl);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            const float u = sigmoidf(up[j] + d.b0);
            st[j] = (1.0f - u) * st[j] + u * tl[j];
        }
        Vec<T>::store(out + r * d.C + c, st);
    }
}

// dtilde = dout * u;  dgates[update half] = dout * (tilde - state) * u * (1 - u)
template <typename T>
__global__ __launch_bounds__(kThreads) void gru_output_bwd_kernel(GruDims d, const T* __restrict__ dout, int ld_dout,
                                                                  const T* __restrict__ gates, const T* __restrict__ xs,
                                                                  const T* __restrict__ tilde, T* __restrict__ dtilde,
                                                                  T* __restrict__ dgates) {
    constexpr int VN = Vec<T>::N;
    const int W = d.Cx + d.C, CV = d.C / VN;
    const int64_t total = d.rows * CV;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / CV;
        const int c = (int)(i % CV) * VN;
        float g[VN], up[VN], st[VN], tl[VN];
        Vec<T>::load(dout + r * ld_dout + c, g);
        Vec<T>::load(gates + r * 2 * d.C + c, up);
        Vec<T>::load(xs + r * W + d.Cx + c, st);
        Vec<T>::load(tilde + r * d.C + c, tl);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            const float u = sigmoidf(up[j] + d.b0);
            up[j] = g[j] * (tl[j] - st[j]) * u * (1.0f - u);
            tl[j] = g[j] * u;
        }
        Vec<T>::store(dtilde + r * d.C + c, tl);
        Vec<T>::store(dgates + r * 2 * d.C + c, up);
    }
}

// acc = [dxs2_x | dout * (1 - u) + dxs2_state * (1 - r)]  (what reaches x and the state NOT through the gate convolution);
// dgates[reset half] = -dxs2_state * state * r * (1 - r)
template <typename T>
__global__ __launch_bounds__(kThreads) void gru_reset_cat_bwd_kernel(GruDims d, const T* __restrict__ dout, int ld_dout,
                                                                     const T* __restrict__ gates, const T* __restrict__ xs,
                                                                     const T* __restrict__ dxs2, T* __restrict__ acc,
                                                                     T* __restrict__ dgates) {
    constexpr int VN = Vec<T>::N;
    const int W = d.Cx + d.C, WV = W / VN;
    const int64_t total = d.rows * WV;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / WV;
        const int c = (int)(i % WV) * VN;
        if (c < d.Cx) {
            Vec<T>::copy(acc + r * W + c, dxs2 + r * W + c);
            continue;
        }
        const int cs = c - d.Cx;
        float g[VN], up[VN], rp[VN], st[VN], dz[VN];
        Vec<T>::load(dout + r * ld_dout + cs, g);
        Vec<T>::load(gates + r * 2 * d.C + cs, up);
        Vec<T>::load(gates + r * 2 * d.C + d.C + cs, rp);
        Vec<T>::load(xs + r * W + c, st);
        Vec<T>::load(dxs2 + r * W + c, dz);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            const float u = sigmoidf(up[j] + d.b0), rg = sigmoidf(rp[j] + d.b0);
            g[j] = g[j] * (1.0f - u) + dz[j] * (1.0f - rg);
            rp[j] = -dz[j] * st[j] * rg * (1.0f - rg);
        }
        Vec<T>::store(acc + r * W + c, g);
        Vec<T>::store(dgates + r * 2 * d.C + d.C + cs, rp);
    }
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline int check(const stp3_gru_dims* p, GruDims* d) {
    if (!p || p->rows <= 0 || p->Cx <= 0 || p->C <= 0) return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    const int vec = p->dtype == STP3_DTYPE_BF16 ? 8 : 4;
    if (p->Cx % vec || p->C % vec) return STP3_EUNSUP;
    d->rows = p->rows; d->Cx = p->Cx; d->C = p->C; d->b0 = p->bias_init;
    return STP3_OK;
}

inline bool aligned16(std::initializer_list<const void*> ps) {
    for (const void* q : ps)
        if ((uintptr_t)q & 15) return false;
    return true;
}

inline unsigned grid_for(int64_t vectors) {
    const int64_t want = (vectors + kThreads - 1) / kThreads;
    const int64_t cap = 256 * 8 * 4;                                     // a few resident rounds, grid-stride beyond
    return (unsigned)(want < cap ? want : cap);
}

}  // namespace

extern "C" {

int stp3_gru_reset_cat_fwd(const stp3_gru_dims* p, const void* xs, const void* gates, void* xs2, void* stream) {
    GruDims d;
    int rc = check(p, &d);
    if (rc) return rc;
    if (!xs || !gates || !xs2) return STP3_EINVAL;
    if (!aligned16({xs, gates, xs2})) return STP3_EUNSUP;
    const bool bf = p->dtype == STP3_DTYPE_BF16;
    const unsigned g = grid_for(d.rows * ((d.Cx + d.C) / (bf ? 8 : 4)));
    hipStream_t s = (hipStream_t)stream;
    if (bf) hipLaunchKernelGGL(gru_reset_cat_fwd_kernel<uint16_t>, dim3(g), dim3(kThreads), 0, s, d, (const uint16_t*)xs,
                               (const uint16_t*)gates, (uint16_t*)xs2);
    else hipLaunchKernelGGL(gru_reset_cat_fwd_kernel<float>, dim3(g), dim3(kThreads), 0, s, d, (const float*)xs,
                            (const float*)gates, (float*)xs2);
    return status();
}

int stp3_gru_output_fwd(const stp3_gru_dims* p, const void* gates, const void* xs, const void* tilde, void* out, void* stream) {
    GruDims d;
    int rc = check(p, &d);
    if (rc) return rc;
    if (!gates || !xs || !tilde || !out) return STP3_EINVAL;
    if (!aligned16({gates, xs, tilde, out})) return STP3_EUNSUP;
    const bool bf = p->dtype == STP3_DTYPE_BF16;
    const unsigned g = grid_for(d.rows * (d.C / (bf ? 8 : 4)));
    hipStream_t s = (hipStream_t)stream;
    if (bf) hipLaunchKernelGGL(gru_output_fwd_kernel<uint16_t>, dim3(g), dim3(kThreads), 0, s, d, (const uint16_t*)gates,
                               (const uint16_t*)xs, (const uint16_t*)tilde, (uint16_t*)out);
    else hipLaunchKernelGGL(gru_output_fwd_kernel<float>, dim3(g), dim3(kThreads), 0, s, d, (const float*)gates,
                            (const float*)xs, (const float*)tilde, (float*)out);
    return status();
}

int stp3_gru_output_bwd(const stp3_gru_dims* p, const void* dout, int32_t ld_dout, const void* gates, const void* xs,
                        const void* tilde, void* dtilde, void* dgates, void* stream) {
    GruDims d;
    int rc = check(p, &d);
    if (rc) return rc;
    if (!dout || !gates || !xs || !tilde || !dtilde || !dgates || ld_dout < p->C) return STP3_EINVAL;
    const bool bf = p->dtype == STP3_DTYPE_BF16;
    if (!aligned16({dout, gates, xs, tilde, dtilde, dgates}) || ld_dout % (bf ? 8 : 4)) return STP3_EUNSUP;
    const unsigned g = grid_for(d.rows * (d.C / (bf ? 8 : 4)));
    hipStream_t s = (hipStream_t)stream;
    if (bf) hipLaunchKernelGGL(gru_output_bwd_kernel<uint16_t>, dim3(g), dim3(kThreads), 0, s, d, (const uint16_t*)dout,
                               ld_dout, (const uint16_t*)gates, (const uint16_t*)xs, (const uint16_t*)tilde,
                               (uint16_t*)dtilde, (uint16_t*)dgates);
    else hipLaunchKernelGGL(gru_output_bwd_kernel<float>, dim3(g), dim3(kThreads), 0, s, d, (const float*)dout, ld_dout,
                            (const float*)gates, (const float*)xs, (const float*)tilde, (float*)dtilde, (float*)dgates);
    return status();
}

int stp3_gru_reset_cat_bwd(const stp3_gru_dims* p, const void* dout, int32_t ld_dout, const void* gates, const void* xs,
                           const void* dxs2, void* acc, void* dgates, void* stream) {
    GruDims d;
    int rc = check(p, &d);
    if (rc) return rc;
    if (!dout || !gates || !xs || !dxs2 || !acc || !dgates || ld_dout < p->C) return STP3_EINVAL;
    const bool bf = p->dtype == STP3_DTYPE_BF16;
    if (!aligned16({dout, gates, xs, dxs2, acc, dgates}) || ld_dout % (bf ? 8 : 4)) return STP3_EUNSUP;
    const unsigned g = grid_for(d.rows * ((d.Cx + d.C) / (bf ? 8 : 4)));
    hipStream_t s = (hipStream_t)stream;
    if (bf) hipLaunchKernelGGL(gru_reset_cat_bwd_kernel<uint16_t>, dim3(g), dim3(kThreads), 0, s, d, (const uint16_t*)dout,
                               ld_dout, (const uint16_t*)gates, (const uint16_t*)xs, (const uint16_t*)dxs2, (uint16_t*)acc,
                               (uint16_t*)dgates);
    else hipLaunchKernelGGL(gru_reset_cat_bwd_kernel<float>, dim3(g), dim3(kThreads), 0, s, d, (const float*)dout, ld_dout,
                            (const float*)gates, (const float*)xs, (const float*)dxs2, (float*)acc, (float*)dgates);
    return status();
}

}  // extern "C"
