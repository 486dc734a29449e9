// stp3_linear.hip -- y = x W^T + b for the POOLED descriptors of the BEV networks, forward and backward in one launch each.
//
// The reference's ASPP image-pooling branch (stp3/layers/convolutions.py:229-240), the pyramid pooling of its temporal blocks
// (stp3/layers/temporal.py:380-424) and the ego-motion planes it concatenates to the BEV (stp3/models/stp3.py:145-152) are
// 1x1 convolutions of tensors that are CONSTANT over the plane; folded, each is a product of a handful of rows -- (12..72) x
// (6..160) by (6..160) x (21..128) -- whose time is launch latency.  torch hands them to hipBLASLt: one launch forward, two
// backward, 9-19 us each, 38 per training step.  Here: one launch each way, float32, deterministic (a fixed 16-lane split of
// every sum, then the DPP row tree of stp3_cdna.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

constexpr int kT = 256;
constexpr int kLanes = 16;                  // lanes that share one output element (one DPP row)

// forward: element e = m * N + n
__global__ __launch_bounds__(kT) void linear_fwd_kernel(int M, int K, int N, const float* __restrict__ x,
                                                        const float* __restrict__ w, int ldw, const float* __restrict__ b,
                                                        float* __restrict__ y) {
    const int lane = threadIdx.x & (kLanes - 1);
    const long e = ((long)blockIdx.x * kT + threadIdx.x) / kLanes;
    const bool ok = e < (long)M * N;
    const long ec = ok ? e : 0;                                   // (all 16 lanes of a row run the tree: clamp, do not exit)
    const int m = (int)(ec / N), n = (int)(ec - (long)m * N);
    const float* xr = x + (size_t)m * K;
    const float* wr = w + (size_t)n * ldw;
    float acc = 0.f;
    for (int k = lane; k < K; k += kLanes) acc = fmaf(xr[k], wr[k], acc);
    acc = row16_sum(acc);
    if (ok && lane == 0) y[e] = acc + (b ? b[n] : 0.f);
}

// backward: elements [0, M*K) = dx, [M*K, M*K + N*K) = dw, then N of db; absent outputs (null) take no elements
//   dx[m][k] = sum_n dy[m][n] w[n][k];   dw[n][k] = sum_m dy[m][n] x[m][k];   db[n] = sum_m dy[m][n]
__global__ __launch_bounds__(kT) void linear_bwd_kernel(int M, int K, int N, const float* __restrict__ dy,
                                                        const float* __restrict__ x, const float* __restrict__ w, int ldw,
                                                        float* __restrict__ dx, float* __restrict__ dw, int lddw,
                                                        float* __restrict__ db) {
    const int lane = threadIdx.x & (kLanes - 1);
    const long n_dx = dx ? (long)M * K : 0, n_dw = dw ? (long)N * K : 0, n_db = db ? N : 0;
    const long e = ((long)blockIdx.x * kT + threadIdx.x) / kLanes;
    const bool ok = e < n_dx + n_dw + n_db;
    float acc = 0.f;
    float* out = nullptr;
    if (ok) {
        if (e < n_dx) {
            const int m = (int)(e / K), k = (int)(e - (long)m * K);
            for (int n = lane; n < N; n += kLanes) acc = fmaf(dy[(size_t)m * N + n], w[(size_t)n * ldw + k], acc);
            out = dx + e;
        } else if (e < n_dx + n_dw) {
            const long q = e - n_dx;
            const int n = (int)(q / K), k = (int)(q - (long)n * K);
            for (int m = lane; m < M; m += kLanes) acc = fmaf(dy[(size_t)m * N + n], x[(size_t)m * K + k], acc);
            out = dw + (size_t)n * lddw + k;
        } else {
            const int n = (int)(e - n_dx - n_dw);
            for (int m = lane; m < M; m += kLanes) acc += dy[(size_t)m * N + n];
            out = db + n;
        }
    }
    acc = row16_sum(acc);                                         // (every lane of the row takes part, valid element or not)
    if (ok && lane == 0) *out = acc;
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

}  // namespace

extern "C" {

int stp3_linear_fwd(int32_t M, int32_t K, int32_t N, const float* x, const float* w, int32_t ldw, const float* b, float* y,
                    void* stream) {
    if (M <= 0 || K <= 0 || N <= 0 || !x || !w || !y || ldw < K) return STP3_EINVAL;
    const int64_t elems = (int64_t)M * N;
    if (elems * kLanes >= (1LL << 31) * (int64_t)kT) return STP3_EUNSUP;
    const unsigned blocks = (unsigned)((elems * kLanes + kT - 1) / kT);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(blocks), dim3(kT), 0, (hipStream_t)stream, (int)M, (int)K, (int)N, x, w, (int)ldw, b, y);
    return status();
}

int stp3_linear_bwd(int32_t M, int32_t K, int32_t N, const float* dy, const float* x, const float* w, int32_t ldw, float* dx,
                    float* dw, int32_t lddw, float* db, void* stream) {
    if (M <= 0 || K <= 0 || N <= 0 || !dy || (dx && (!w || ldw < K)) || (dw && (!x || lddw < K))) return STP3_EINVAL;
    const int64_t elems = (dx ? (int64_t)M * K : 0) + (dw ? (int64_t)N * K : 0) + (db ? N : 0);
    if (elems == 0) return STP3_OK;
    if (elems * kLanes >= (1LL << 31) * (int64_t)kT) return STP3_EUNSUP;
    const unsigned blocks = (unsigned)((elems * kLanes + kT - 1) / kT);
    hipLaunchKernelGGL(linear_bwd_kernel, dim3(blocks), dim3(kT), 0, (hipStream_t)stream, (int)M, (int)K, (int)N, dy, x, w, (int)ldw, dx, dw, (int)lddw, db);
    return status();
}

}  // extern "C"
