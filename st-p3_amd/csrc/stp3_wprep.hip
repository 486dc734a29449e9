// stp3_wprep.hip -- one launch that refreshes the bf16 shadow copies of EVERY convolution weight (gfx950).
//
// The dense-convolution kernels of stp3_conv.hip read bf16 weights in [Cout][KH][KW][Cin] order and, for the data
// gradient, the tap-flipped / channel-swapped copy [Cin][KH][KW][Cout].  The fp32 master weights live in the flat
// parameter buckets the optimizer updates (stp3_amd/parallel.py), so after every optimizer step each of the ~70
// layers would need a cast, a flip, a transpose and a re-layout -- several hundred tiny launches.  Here a table in
// device memory describes all layers (pointers are stable: the parameters are views of the flat buckets) and one
// kernel rewrites all shadows; the convolution operators then never touch torch for their weights.  Weights the model ASSEMBLES from
// parameters (zero-padded channel lanes, the taps of a causal 3-D kernel side by side, merged heads, a split projection) are
// written piece by piece by the same launch, and their gradients go back to the parameters in one launch too.
//
// Work split: the table carries an exclusive scan of 256-element blocks per layer; a workgroup binary-searches
// its layer.  Reads follow the forward layout (coalesced for channels-last masters), the flipped copy is written
// with a Cout stride -- the whole model is ~8 M weights, far below anything that matters for HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

__device__ __forceinline__ uint16_t f2bf(float a) {   // round to nearest even (== torch .to(bfloat16))
    uint32_t u = __float_as_uint(a);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// the table entry whose block range contains this workgroup: last entry with first_block <= blockIdx.x
__device__ __forceinline__ int find_entry(const stp3_wprep_entry* __restrict__ table, int n, int64_t b) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One PIECE per entry: a (cout, cin, kh, kw) block read through arbitrary strides (a parameter, or a view of one: a tap of a
// 3-D kernel, a channel split, the centre of a 3x3) and written at the channel offsets (co_off, ci_off) of a destination of
// dst_cout x dst_cin channels -- a weight ASSEMBLED from several parameters (zero lanes in between are never written: the
// host zeroes the destination once).  dst_cout == 0: the piece is the whole weight.
__global__ __launch_bounds__(256) void prep_weights_kernel(const stp3_wprep_entry* __restrict__ table, int n) {
    const int64_t b = blockIdx.x;
    const stp3_wprep_entry e = table[find_entry(table, n, b)];
    const int64_t total = (int64_t)e.cout * e.cin * e.kh * e.kw;
    const int64_t i = (b - e.first_block) * 256 + threadIdx.x;   // index in the piece's forward layout [co][r][s][ci]
    if (i >= total) return;
    const int ci = (int)(i % e.cin);
    int64_t t = i / e.cin;
    const int s = (int)(t % e.kw);
    t /= e.kw;
    const int r = (int)(t % e.kh);
    const int co = (int)(t / e.kh);
    const float v = e.src[co * e.stride_co + ci * e.stride_ci + r * e.stride_kh + s * e.stride_kw];
    const int dco = e.dst_cout ? e.dst_cout : e.cout, dci = e.dst_cout ? e.dst_cin : e.cin;
    if (e.fwd) {
        const int64_t k = ((((int64_t)co + e.co_off) * e.kh + r) * e.kw + s) * dci + ci + e.ci_off;
        if (e.fwd_f32) ((float*)e.fwd)[k] = v;                   // (the depthwise kernels read float32 taps)
        else ((uint16_t*)e.fwd)[k] = f2bf(v);
    }
    if (e.flip) {
        const int64_t j = ((((int64_t)ci + e.ci_off) * e.kh + (e.kh - 1 - r)) * e.kw + (e.kw - 1 - s)) * dco + co + e.co_off;
        ((uint16_t*)e.flip)[j] = f2bf(v);
    }
}

// The way back for the GRADIENT of an assembled weight: every piece of the float32 gradient [dst_cout][KH][KW][dst_cin] (what
// stp3_conv2d_wgrad writes) goes to its parameter's gradient -- ``src`` is that destination here, through the same strides.
__global__ __launch_bounds__(256) void scatter_weight_grads_kernel(const stp3_wprep_entry* __restrict__ table, int n) {
    const int64_t b = blockIdx.x;
    const stp3_wprep_entry e = table[find_entry(table, n, b)];
    const int64_t total = (int64_t)e.cout * e.cin * e.kh * e.kw;
    const int64_t i = (b - e.first_block) * 256 + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % e.cin);
    int64_t t = i / e.cin;
    const int s = (int)(t % e.kw);
    t /= e.kw;
    const int r = (int)(t % e.kh);
    const int co = (int)(t / e.kh);
    const int dci = e.dst_cout ? e.dst_cin : e.cin;
    const int64_t k = ((((int64_t)co + e.co_off) * e.kh + r) * e.kw + s) * dci + ci + e.ci_off;
    ((float*)e.src)[co * e.stride_co + ci * e.stride_ci + r * e.stride_kh + s * e.stride_kw] = ((const float*)e.fwd)[k];
}

}  // namespace

extern "C" {

int stp3_conv2d_prep_weights(const stp3_wprep_entry* table, int32_t n_entries, int64_t total_blocks, void* stream) {
    if (n_entries < 0 || total_blocks < 0) return STP3_EINVAL;
    if (n_entries == 0 || total_blocks == 0) return STP3_OK;
    if (!table) return STP3_EINVAL;
    if (total_blocks >= (1LL << 31)) return STP3_EUNSUP;
    hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table,
                       n_entries);
    return launch_status();
}

int stp3_conv2d_scatter_weight_grads(const stp3_wprep_entry* table, int32_t n_entries, int64_t total_blocks, void* stream) {
    if (n_entries < 0 || total_blocks < 0) return STP3_EINVAL;
    if (n_entries == 0 || total_blocks == 0) return STP3_OK;
    if (!table) return STP3_EINVAL;
    if (total_blocks >= (1LL << 31)) return STP3_EUNSUP;
    hipLaunchKernelGGL(scatter_weight_grads_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table,
                       n_entries);
    return launch_status();
}

}  // extern "C"
