// stp3_optim.hip -- gradient-norm clipping + Adam over the flat parameter buckets in three launches (gfx950).
//
// The training step of the reference ends with gradient_clip_val (train.py:48) and torch.optim.Adam
// (trainer.py:456-462).  On flat fp32 buckets (stp3_amd/parallel.py) that is, written with torch operators, about
// a dozen elementwise passes and ~20 launches per bucket.  Here:
//   1. optim_sumsq_kernel     per-workgroup partial sums of g^2 over ALL buckets (fixed work split -> deterministic)
//   2. optim_prepare_kernel   one workgroup: total norm, clip scale, step counter += 1, Adam bias corrections
//   3. optim_adam_kernel      g <- g * scale (written back, like clip_grad_norm_), weight decay, moments, update
// Everything stays on the device (no host sync, graph-capturable); HBM-bound: step 3 reads g, p, m, v and writes
// g, p, m, v = 32 B per parameter (8.35 M parameters: 267 MB, ~45 us at 6 TB/s).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

constexpr int kChunk = 256 * 16;   // elements per workgroup: 256 lanes x 4 float4

// the bucket whose block range contains workgroup b: last entry with first_block <= b
__device__ __forceinline__ int find_bucket(const stp3_optim_bucket* __restrict__ t, int n, int64_t b) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void optim_sumsq_kernel(const stp3_optim_bucket* __restrict__ table, int n,
                                                          float* __restrict__ partial) {
    __shared__ float red[256];
    const stp3_optim_bucket e = table[find_bucket(table, n, blockIdx.x)];
    const int64_t base = ((int64_t)blockIdx.x - e.first_block) * kChunk;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < e.numel) {
            const float g = e.grad[i];
            s = fmaf(g, g, s);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// state: [0] step count, [1] clip scale, [2] lr / (1 - b1^t), [3] sqrt(1 - b2^t), [4] total gradient norm
__global__ __launch_bounds__(256) void optim_prepare_kernel(int64_t n_partial, const float* __restrict__ partial,
                                                            float max_norm, float lr, float beta1, float beta2,
                                                            float* __restrict__ state) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n_partial; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0]);
        const float t = state[0] + 1.0f;
        state[0] = t;
        float scale = 1.0f;
        if (max_norm > 0.f) scale = fminf(max_norm / (total + 1e-6f), 1.0f);   // torch clip_grad_norm_
        state[1] = scale;
        state[2] = lr / (1.0f - powf(beta1, t));
        state[3] = sqrtf(1.0f - powf(beta2, t));
        state[4] = total;
    }
}

__global__ __launch_bounds__(256) void optim_adam_kernel(const stp3_optim_bucket* __restrict__ table, int n,
                                                         const float* __restrict__ state, float beta1, float beta2,
                                                         float eps, float weight_decay) {
    const stp3_optim_bucket e = table[find_bucket(table, n, blockIdx.x)];
    const int64_t base = ((int64_t)blockIdx.x - e.first_block) * kChunk;
    const float scale = state[1], step_size = state[2], bc2_sqrt = state[3];
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i >= e.numel) break;
        const float gc = e.grad[i] * scale;
        const float p = e.param[i];
        const float g = weight_decay != 0.f ? fmaf(weight_decay, p, gc) : gc;   // L2 decay folded into the gradient
        float m = e.exp_avg[i], v = e.exp_avg_sq[i];
        m = fmaf(1.0f - beta1, g - m, m);                                        // lerp
        v = fmaf((1.0f - beta2) * g, g, v * beta2);
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        e.grad[i] = gc;
        e.exp_avg[i] = m;
        e.exp_avg_sq[i] = v;
        e.param[i] = p - (m / denom) * step_size;
    }
}

}  // namespace

extern "C" {

int stp3_optim_workspace_bytes(int64_t total_blocks, size_t* bytes) {
    if (total_blocks < 0 || !bytes) return STP3_EINVAL;
    *bytes = (size_t)(total_blocks > 0 ? total_blocks : 1) * sizeof(float);
    return STP3_OK;
}

int stp3_optim_clip_adam(const stp3_optim_bucket* table, int32_t n_buckets, int64_t total_blocks, float max_norm,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float* state,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (n_buckets < 0 || total_blocks < 0) return STP3_EINVAL;
    if (n_buckets == 0 || total_blocks == 0) return STP3_OK;
    if (!table || !state || !workspace) return STP3_EINVAL;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f)) return STP3_EINVAL;
    if (total_blocks >= (1LL << 31)) return STP3_EUNSUP;
    if (workspace_bytes < (size_t)total_blocks * sizeof(float)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(optim_sumsq_kernel, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_buckets, partial);
    hipLaunchKernelGGL(optim_prepare_kernel, dim3(1), dim3(256), 0, s, total_blocks, partial, max_norm, lr, beta1, beta2,
                       state);
    hipLaunchKernelGGL(optim_adam_kernel, dim3((unsigned)total_blocks), dim3(256), 0, s, table, n_buckets, state, beta1,
                       beta2, eps, weight_decay);
    return launch_status();
}

}  // extern "C"
