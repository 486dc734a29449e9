// stp3_temporal.hip -- the operand of the causal (2,3,3) convolution of the temporal model, for gfx950.
//
// The reference's CausalConv3d (stp3/layers/temporal.py:252-273) pads time on the left by one frame, so that
// y[t] = W[:, :, 0] * x[t-1] + W[:, :, 1] * x[t] with x[-1] = 0.  On the frame-folded channels-last sequence that is ONE
// 2-D 3x3 convolution over the channel pairing [x[t-1], x[t]].  Built with torch operators the pairing costs a zero
// frame, two concatenations and, in the backward pass, a zero-filled slice gradient plus an accumulation; here it is
// one streaming pass each way (16-byte channel vectors; HBM-bound: reads C, writes 2C lanes per pixel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

constexpr int kThreads = 256;

struct PairDims {
    int frames, T, rows, CV, ldv;     // CV: 16-byte vectors per C channels; ldv: row stride of x in vectors
};

// y [frames][rows][2 CV]: first CV vectors = the previous frame of the same sample (zero for its first frame), last CV
// vectors = this frame
__global__ __launch_bounds__(kThreads) void causal_pair_fwd_kernel(PairDims d, const uint4* __restrict__ x,
                                                                   uint4* __restrict__ y, long total) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int v2 = 2 * d.CV;
    const long row = i / v2;                       // global row: frame * rows + r
    const int v = (int)(i - row * v2);
    const int n = (int)(row / d.rows);
    const bool cur = v >= d.CV;
    uint4 val = make_uint4(0u, 0u, 0u, 0u);
    if (cur || n % d.T != 0) {
        const long src_row = cur ? row : row - d.rows;
        val = x[src_row * d.ldv + (cur ? v - d.CV : v)];
    }
    y[i] = val;
}

template <bool BF16>
__device__ __forceinline__ uint4 add_vec(uint4 a, uint4 b) {
    if (BF16) {
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = __uint_as_float(aw[k] << 16) + __uint_as_float(bw[k] << 16);
            const float hi = __uint_as_float(aw[k] & 0xffff0000u) + __uint_as_float(bw[k] & 0xffff0000u);
            o[k] = pack_bf16(lo, hi);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
    return make_uint4(__float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)),
                      __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y)),
                      __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)),
                      __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w)));
}

// dx [frames][rows][CV] = dy[n][r][CV + v] + (dy[n+1][r][v] unless n is the last frame of its sample)
template <bool BF16>
__global__ __launch_bounds__(kThreads) void causal_pair_bwd_kernel(PairDims d, const uint4* __restrict__ dy,
                                                                   uint4* __restrict__ dx, long total) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const long row = i / d.CV;
    const int v = (int)(i - row * d.CV);
    const int n = (int)(row / d.rows);
    const int v2 = 2 * d.CV;
    uint4 g = dy[row * v2 + d.CV + v];
    if (n % d.T != d.T - 1) g = add_vec<BF16>(g, dy[(row + d.rows) * v2 + v]);
    dx[i] = g;
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline int plan(const stp3_pair_dims* p, PairDims* d, std::initializer_list<const void*> ptrs) {
    if (!p) return STP3_EINVAL;
    if (p->frames <= 0 || p->T <= 0 || p->frames % p->T != 0 || p->rows <= 0 || p->C <= 0 || p->ldx < p->C)
        return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    const int per = p->dtype == STP3_DTYPE_BF16 ? 8 : 4;          // elements per 16-byte vector
    if (p->C % per != 0 || p->ldx % per != 0) return STP3_EUNSUP;
    for (const void* q : ptrs) {
        if (!q) return STP3_EINVAL;
        if ((uintptr_t)q & 15) return STP3_EUNSUP;
    }
    d->frames = p->frames; d->T = p->T; d->rows = p->rows; d->CV = p->C / per; d->ldv = p->ldx / per;
    return STP3_OK;
}

}  // namespace

extern "C" {

int stp3_causal_pair_fwd(const stp3_pair_dims* p, const void* x, void* y, void* stream) {
    PairDims d;
    const int rc = plan(p, &d, {x, y});
    if (rc != STP3_OK) return rc;
    const long total = (long)d.frames * d.rows * 2 * d.CV;
    hipLaunchKernelGGL(causal_pair_fwd_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, d, (const uint4*)x, (uint4*)y, total);
    return status();
}

int stp3_causal_pair_bwd(const stp3_pair_dims* p, const void* dy, void* dx, void* stream) {
    PairDims d;
    const int rc = plan(p, &d, {dy, dx});
    if (rc != STP3_OK) return rc;
    const long total = (long)d.frames * d.rows * d.CV;
    const dim3 grid((unsigned)((total + kThreads - 1) / kThreads));
    if (p->dtype == STP3_DTYPE_BF16)
        hipLaunchKernelGGL(causal_pair_bwd_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, d,
                           (const uint4*)dy, (uint4*)dx, total);
    else
        hipLaunchKernelGGL(causal_pair_bwd_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, d,
                           (const uint4*)dy, (uint4*)dx, total);
    return status();
}

}  // extern "C"
