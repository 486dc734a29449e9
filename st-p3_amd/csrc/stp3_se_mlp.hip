// stp3_se_mlp.hip -- the two tiny fully-connected layers of a squeeze-and-excitation block as single launches.
//
// An MBConv block's gate is  sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2)  on (N, C) / (N, S) tensors with
// N = B*T*cameras = 72, C <= 960, S <= 40 in the EfficientNet-B4 trunk (stp3/models/encoder.py:57-97 drives
// efficientnet_pytorch's MBConvBlock).  Written with torch operators that is ~10 launches forward and ~15 backward
// per block, 22 blocks per step, every one of them launch-latency bound.  Here: one launch forward (a workgroup per
// sample), two backward (per-sample chain, then the weight gradients reduced over the samples in a fixed order).
// EXPERIMENTAL: host side selected with STP3_SE_MLP=1 on top of STP3_FUSED_SE=1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

__device__ __forceinline__ float wave_sum_xor(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

constexpr int kRows = 5;        // rows of the small weight matrices a wave reduces together (S <= 40: two rounds)

// forward: grid = N, LDS = (C + S) floats
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(stp3_se_mlp_dims d, const float* __restrict__ pooled_sum,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ z1, float* __restrict__ gate) {
    extern __shared__ float smem[];
    float* p = smem;            // [C] pooled mean
    float* h = smem + d.C;      // [S] swish(z1)
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < d.C; c += 256) p[c] = pooled_sum[(size_t)n * d.C + c] * d.inv_rows;
    __syncthreads();
    // rows s = wave, wave + 4, ... of W1, kRows at a time: the loads of kRows rows are in flight together and their
    // wave reductions interleave (one row at a time was a chain of S / 4 dependent load -> reduce round trips: 27 us)
    for (int s0 = wave; s0 < d.S; s0 += 4 * kRows) {
        float acc[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) acc[k] = 0.f;
#pragma unroll 4
        for (int c = lane; c < d.C; c += 64) {            // (unrolled: the loads of four iterations in flight)
            const float pc = p[c];
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int s = s0 + 4 * k;
                if (s < d.S) acc[k] = fmaf(w1[(size_t)s * d.C + c], pc, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k) acc[k] = wave_sum_xor(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int s = s0 + 4 * k;
                if (s < d.S) {
                    const float z = acc[k] + b1[s];
                    z1[(size_t)n * d.S + s] = z;
                    h[s] = z * sigmoidf_(z);
                }
            }
        }
    }
    __syncthreads();
    // a thread owns channel c and walks ITS row of W2 (S consecutive floats): 16-byte loads when the row length allows
    // (S = 8, 12, 28, 40 of EfficientNet-B4; 6 and 14 take 8-byte loads) -- lane-to-lane the rows are S * 4 bytes apart, so
    // every load instruction touches 64 cache lines whatever its width: four times fewer of them
    for (int c = tid; c < d.C; c += 256) {
        float a = b2[c];
        const float* row = w2 + (size_t)c * d.S;
        if ((d.S & 3) == 0 && ((uintptr_t)w2 & 15) == 0) {
            for (int s = 0; s < d.S; s += 4) {
                const float4 w = *reinterpret_cast<const float4*>(row + s);
                a = fmaf(w.x, h[s], a); a = fmaf(w.y, h[s + 1], a); a = fmaf(w.z, h[s + 2], a); a = fmaf(w.w, h[s + 3], a);
            }
        } else if ((d.S & 1) == 0 && ((uintptr_t)w2 & 7) == 0) {
            for (int s = 0; s < d.S; s += 2) {
                const float2 w = *reinterpret_cast<const float2*>(row + s);
                a = fmaf(w.x, h[s], a); a = fmaf(w.y, h[s + 1], a);
            }
        } else {
            for (int s = 0; s < d.S; ++s) a = fmaf(row[s], h[s], a);
        }
        gate[(size_t)n * d.C + c] = sigmoidf_(a);
    }
}

// backward, per sample: dz2 = dgate * gate * (1 - gate); dh = dz2 W2; dz1 = dh * swish'(z1); dpooled = dz1 W1 / rows
// grid = N, LDS = (C + 2 S) floats
__global__ __launch_bounds__(256) void se_mlp_bwd_sample_kernel(stp3_se_mlp_dims d, const float* __restrict__ dgate,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ z1,
                                                                const float* __restrict__ w1,
                                                                const float* __restrict__ w2, float* __restrict__ dz2,
                                                                float* __restrict__ dz1, float* __restrict__ dpooled) {
    extern __shared__ float smem[];
    float* g2 = smem;               // [C] dz2 of this sample
    float* g1 = smem + d.C;         // [S] dz1 of this sample
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < d.C; c += 256) {
        const float g = gate[(size_t)n * d.C + c];
        const float v = dgate[(size_t)n * d.C + c] * g * (1.0f - g);
        g2[c] = v;
        dz2[(size_t)n * d.C + c] = v;
    }
    __syncthreads();
    for (int s0 = wave; s0 < d.S; s0 += 4 * kRows) {            // kRows columns of W2 at a time (see the forward kernel)
        float acc[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) acc[k] = 0.f;
#pragma unroll 4
        for (int c = lane; c < d.C; c += 64) {
            const float gc = g2[c];
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int s = s0 + 4 * k;
                if (s < d.S) acc[k] = fmaf(gc, w2[(size_t)c * d.S + s], acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k) acc[k] = wave_sum_xor(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int s = s0 + 4 * k;
                if (s < d.S) {
                    const float z = z1[(size_t)n * d.S + s];
                    const float sg = sigmoidf_(z);
                    const float v = acc[k] * sg * (1.0f + z * (1.0f - sg));
                    g1[s] = v;
                    dz1[(size_t)n * d.S + s] = v;
                }
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < d.C; c += 256) {
        float a = 0.f;
#pragma unroll 8
        for (int s = 0; s < d.S; ++s) a = fmaf(g1[s], w1[(size_t)s * d.C + c], a);
        dpooled[(size_t)n * d.C + c] = a * d.inv_rows;
    }
}

// backward, weights: thread = channel c, workgroup = (64 channels, 8 squeezed channels); samples added in ascending
// order (deterministic)
//   dw2[c][s] = sum_n dz2[n][c] swish(z1[n][s]);  db2[c] = sum_n dz2[n][c]
//   dw1[s][c] = sum_n dz1[n][s] pooled[n][c];      db1[s] = sum_n dz1[n][s]   (workgroup (0, 0))
// LDS = 2 * N * 8 floats (swish(z1) and dz1 of all samples for the workgroup's 8 squeezed channels).  The grid is
// C/64 x S/8 workgroups: the first version ran C/256 workgroups (1-4 on the whole chip) for 78 us per call.
constexpr int kSChunk = 8;
constexpr int kWgtThreads = 64;

__global__ __launch_bounds__(kWgtThreads) void se_mlp_bwd_weight_kernel(stp3_se_mlp_dims d,
                                                                        const float* __restrict__ pooled_sum,
                                                                        const float* __restrict__ z1,
                                                                        const float* __restrict__ dz2,
                                                                        const float* __restrict__ dz1,
                                                                        float* __restrict__ dw1, float* __restrict__ db1,
                                                                        float* __restrict__ dw2, float* __restrict__ db2) {
    extern __shared__ float smem[];
    float* hs = smem;                         // [N][8]
    float* g1 = smem + (size_t)d.N * kSChunk; // [N][8]
    const int tid = threadIdx.x;
    const int s0 = blockIdx.y * kSChunk;
    for (int i = tid; i < d.N * kSChunk; i += kWgtThreads) {
        const int n = i / kSChunk, s = s0 + i % kSChunk;
        const float z = s < d.S ? z1[n * d.S + s] : 0.f;
        hs[i] = z * sigmoidf_(z);
        g1[i] = s < d.S ? dz1[n * d.S + s] : 0.f;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < kSChunk && s0 + tid < d.S) {
        float a = 0.f;
        for (int n = 0; n < d.N; ++n) a += g1[n * kSChunk + tid];
        db1[s0 + tid] = a;
    }
    const int c = blockIdx.x * kWgtThreads + tid;
    if (c >= d.C) return;
    float sb = 0.f;
    float a2[kSChunk], a1[kSChunk];
#pragma unroll
    for (int k = 0; k < kSChunk; ++k) a2[k] = a1[k] = 0.f;
#pragma unroll 4
    for (int n = 0; n < d.N; ++n) {
        const float x2 = dz2[(size_t)n * d.C + c];
        const float pc = pooled_sum[(size_t)n * d.C + c] * d.inv_rows;
        sb += x2;
#pragma unroll
        for (int k = 0; k < kSChunk; ++k) {
            a2[k] = fmaf(x2, hs[n * kSChunk + k], a2[k]);
            a1[k] = fmaf(g1[n * kSChunk + k], pc, a1[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < kSChunk; ++k) {
        const int s = s0 + k;
        if (s < d.S) {
            dw2[(size_t)c * d.S + s] = a2[k];
            dw1[(size_t)s * d.C + c] = a1[k];
        }
    }
    if (blockIdx.y == 0) db2[c] = sb;
}

inline int check(const stp3_se_mlp_dims* d) {
    if (!d) return STP3_EINVAL;
    if (d->N <= 0 || d->C <= 0 || d->S <= 0) return STP3_EINVAL;
    if ((size_t)(d->C + 2 * d->S) * 4 > 60 * 1024 || (size_t)2 * d->N * d->S * 4 > 60 * 1024) return STP3_EUNSUP;
    return STP3_OK;
}

}  // namespace

extern "C" {

int stp3_se_mlp_fwd(const stp3_se_mlp_dims* dims, const float* pooled_sum, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* z1, float* gate, void* stream) {
    int rc = check(dims);
    if (rc) return rc;
    if (!pooled_sum || !w1 || !b1 || !w2 || !b2 || !z1 || !gate) return STP3_EINVAL;
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(dims->N), dim3(256), (size_t)(dims->C + dims->S) * 4, (hipStream_t)stream,
                       *dims, pooled_sum, w1, b1, w2, b2, z1, gate);
    return launch_status();
}

int stp3_se_mlp_bwd(const stp3_se_mlp_dims* dims, const float* dgate, const float* gate, const float* pooled_sum,
                    const float* z1, const float* w1, const float* w2, float* dz2, float* dz1, float* dpooled,
                    float* dw1, float* db1, float* dw2, float* db2, void* stream) {
    int rc = check(dims);
    if (rc) return rc;
    if (!dgate || !gate || !pooled_sum || !z1 || !w1 || !w2 || !dz2 || !dz1 || !dpooled || !dw1 || !db1 || !dw2 || !db2)
        return STP3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(se_mlp_bwd_sample_kernel, dim3(dims->N), dim3(256), (size_t)(dims->C + 2 * dims->S) * 4, s, *dims,
                       dgate, gate, z1, w1, w2, dz2, dz1, dpooled);
    hipLaunchKernelGGL(se_mlp_bwd_weight_kernel, dim3((dims->C + kWgtThreads - 1) / kWgtThreads, (dims->S + kSChunk - 1) / kSChunk),
                       dim3(kWgtThreads), (size_t)2 * dims->N * kSChunk * 4, s, *dims, pooled_sum, z1, dz2, dz1, dw1, db1,
                       dw2, db2);
    return launch_status();
}

}  // extern "C"
