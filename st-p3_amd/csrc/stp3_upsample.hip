// stp3_upsample.hip -- bilinear up-sampling (align_corners = False, integer scale) of channels-last maps, forward and
// backward, for gfx950.
//
// Replaces nn.Upsample(scale_factor=2, mode='bilinear') of the reference's hot path: UpsamplingConcat and UpsamplingAdd
// (stp3/layers/convolutions.py:183-215), i.e. the encoder's 14x30 -> 28x60 merge (stp3/models/encoder.py:57-97) and the
// three decoder stages 25 -> 50 -> 100 -> 200 (stp3/models/decoder.py:22-140).  Under autocast torch runs these in
// float32 (cast in, interpolate, cast out; the backward of the 12 x 64 x 200 x 200 stage alone took 0.38 ms); here one
// pass each way over 16-byte channel vectors of the bf16 (or float32) tensors, float32 arithmetic in the same order
// as torch's kernel, one rounding.  HBM-bound: forward reads 1, writes s^2 pixel vectors; backward the reverse.
//   forward : y[oh][ow] = l0h * (l0w * x[h0][w0] + l1w * x[h0][w1]) + l1h * (l0w * x[h1][w0] + l1w * x[h1][w1]),
//             src = max((o + 0.5) / s - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, size - 1), l1 = src - i0, l0 = 1 - l1
//   backward: a gather -- every input pixel sums the <= 2s x 2s output gradients that read it (deterministic, no atomics)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxScale = 4;

struct UpDims {
    int N, H, W, CV, S, Ho, Wo, ldxv, ldyv;   // CV / ld*: in 16-byte vectors
    float inv;                                // 1 / S as torch rounds it
};

template <bool BF16> struct Vec;
template <> struct Vec<true> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(uint4 v, float* f) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
    }
};
template <> struct Vec<false> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(uint4 v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};

// source index and weights of output coordinate o along an axis of `size` input pixels
__device__ __forceinline__ void source(int o, float inv, int size, int* i0, int* i1, float* l0, float* l1) {
    const float src = fmaxf(inv * ((float)o + 0.5f) - 0.5f, 0.f);
    const int a = (int)src;
    *i0 = a;
    *i1 = a + (a < size - 1 ? 1 : 0);
    *l1 = src - (float)a;
    *l0 = 1.f - *l1;
}

template <bool BF16>
__global__ __launch_bounds__(kThreads) void upsample_fwd_kernel(UpDims d, const uint4* __restrict__ x,
                                                                uint4* __restrict__ y, int total) {
    constexpr int V = Vec<BF16>::N;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int cv = i % d.CV;
    int p = i / d.CV;
    const int ow = p % d.Wo;
    p /= d.Wo;
    const int oh = p % d.Ho;
    const int n = p / d.Ho;
    int h0, h1, w0, w1;
    float l0h, l1h, l0w, l1w;
    source(oh, d.inv, d.H, &h0, &h1, &l0h, &l1h);
    source(ow, d.inv, d.W, &w0, &w1, &l0w, &l1w);
    const uint4* xs = x + (size_t)n * d.H * d.W * d.ldxv + cv;
    float a[V], b[V], c[V], e[V], o[V];
    Vec<BF16>::unpack(xs[(size_t)(h0 * d.W + w0) * d.ldxv], a);
    Vec<BF16>::unpack(xs[(size_t)(h0 * d.W + w1) * d.ldxv], b);
    Vec<BF16>::unpack(xs[(size_t)(h1 * d.W + w0) * d.ldxv], c);
    Vec<BF16>::unpack(xs[(size_t)(h1 * d.W + w1) * d.ldxv], e);
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = l0h * (l0w * a[j] + l1w * b[j]) + l1h * (l0w * c[j] + l1w * e[j]);
    y[((size_t)(n * d.Ho + oh) * d.Wo + ow) * d.ldyv + cv] = Vec<BF16>::pack(o);
}

// weights with which the outputs lo .. lo + cnt - 1 along one axis read input pixel `i`
__device__ __forceinline__ void gather_weights(int i, int S, float inv, int size, int out_size, int* lo, int* cnt,
                                               float* wgt) {
    // o reads i when floor(src) == i or floor(src) + 1 == i, src in (i - 1, i + 1): 2o + 1 in (S (2i - 1), S (2i + 3))
    int first = (S * (2 * i - 1)) >> 1;            // a superset by at most one on either side: weights of 0 there
    if (first < 0) first = 0;
    int last = (S * (2 * i + 3)) >> 1;
    if (last > out_size - 1) last = out_size - 1;
    *lo = first;
    *cnt = last - first + 1;
    for (int k = 0; k < *cnt; ++k) {
        int i0, i1;
        float l0, l1;
        source(first + k, inv, size, &i0, &i1, &l0, &l1);
        wgt[k] = (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
    }
}

template <bool BF16>
__global__ __launch_bounds__(kThreads) void upsample_bwd_kernel(UpDims d, const uint4* __restrict__ dy,
                                                                uint4* __restrict__ dx, int total) {
    constexpr int V = Vec<BF16>::N;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int cv = i % d.CV;
    int p = i / d.CV;
    const int iw = p % d.W;
    p /= d.W;
    const int ih = p % d.H;
    const int n = p / d.H;
    int oh0, nh, ow0, nw;
    float wh[2 * kMaxScale + 2], ww[2 * kMaxScale + 2];
    gather_weights(ih, d.S, d.inv, d.H, d.Ho, &oh0, &nh, wh);
    gather_weights(iw, d.S, d.inv, d.W, d.Wo, &ow0, &nw, ww);
    const uint4* gs = dy + (size_t)n * d.Ho * d.Wo * d.ldyv + cv;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int a = 0; a < nh; ++a) {
        if (wh[a] == 0.f) continue;
        for (int b = 0; b < nw; ++b) {
            const float wgt = wh[a] * ww[b];
            if (wgt == 0.f) continue;
            float g[V];
            Vec<BF16>::unpack(gs[(size_t)((oh0 + a) * d.Wo + ow0 + b) * d.ldyv], g);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = fmaf(wgt, g[j], acc[j]);
        }
    }
    dx[((size_t)(n * d.H + ih) * d.W + iw) * d.ldxv + cv] = Vec<BF16>::pack(acc);
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline int plan(const stp3_upsample_dims* p, UpDims* d, std::initializer_list<const void*> ptrs) {
    if (!p) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->C <= 0 || p->scale < 1 || p->ldx < p->C || p->ldy < p->C)
        return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (p->scale > kMaxScale) return STP3_EUNSUP;
    const int per = p->dtype == STP3_DTYPE_BF16 ? 8 : 4;
    if (p->C % per != 0 || p->ldx % per != 0 || p->ldy % per != 0) return STP3_EUNSUP;
    for (const void* q : ptrs) {
        if (!q) return STP3_EINVAL;
        if ((uintptr_t)q & 15) return STP3_EUNSUP;
    }
    const int64_t out_vecs = (int64_t)p->N * p->H * p->scale * p->W * p->scale * (p->ldy / per);
    const int64_t in_vecs = (int64_t)p->N * p->H * p->W * (p->ldx / per);
    if (out_vecs >= (1LL << 31) || in_vecs >= (1LL << 31)) return STP3_EUNSUP;
    d->N = p->N; d->H = p->H; d->W = p->W; d->CV = p->C / per; d->S = p->scale;
    d->Ho = p->H * p->scale; d->Wo = p->W * p->scale; d->ldxv = p->ldx / per; d->ldyv = p->ldy / per;
    d->inv = (float)(1.0 / (double)p->scale);
    return STP3_OK;
}

}  // namespace

extern "C" {

int stp3_upsample_bilinear_fwd(const stp3_upsample_dims* p, const void* x, void* y, void* stream) {
    UpDims d;
    const int rc = plan(p, &d, {x, y});
    if (rc != STP3_OK) return rc;
    const int total = d.N * d.Ho * d.Wo * d.CV;
    const dim3 grid((total + kThreads - 1) / kThreads);
    if (p->dtype == STP3_DTYPE_BF16)
        hipLaunchKernelGGL(upsample_fwd_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, d, (const uint4*)x,
                           (uint4*)y, total);
    else
        hipLaunchKernelGGL(upsample_fwd_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, d, (const uint4*)x,
                           (uint4*)y, total);
    return status();
}

int stp3_upsample_bilinear_bwd(const stp3_upsample_dims* p, const void* dy, void* dx, void* stream) {
    UpDims d;
    const int rc = plan(p, &d, {dy, dx});
    if (rc != STP3_OK) return rc;
    const int total = d.N * d.H * d.W * d.CV;
    const dim3 grid((total + kThreads - 1) / kThreads);
    if (p->dtype == STP3_DTYPE_BF16)
        hipLaunchKernelGGL(upsample_bwd_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, d, (const uint4*)dy,
                           (uint4*)dx, total);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, d, (const uint4*)dy,
                           (uint4*)dx, total);
    return status();
}

}  // extern "C"
