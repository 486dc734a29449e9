// stp3_voxsum.hip -- gfx950 kernels + C ABI for the stand-alone voxel summing operator.
//
// Operator-level twin of the reference's VoxelsSumming (stp3/utils/geometry.py:299-330): rows of x
// that are already sorted by voxel rank are summed per voxel (forward) and the per-voxel gradient
// is handed back to every row of the voxel (backward).  The fused lift path (stp3_lift.hip) never
// builds the sorted row matrix, so this file exists for callers that still hold one -- the
// compatibility shim stp3_amd.geometry.VoxelsSumming.
//
// Where the reference takes a running sum over ALL rows and differences it at the voxel
// boundaries (float32 cancellation grows with the row index), a segment is summed here on its
// own, rows in ascending order, so the result does not depend on what precedes the voxel.
//
// Layout: x [M][C] float32 row-major, seg_off [S+1] int32 ascending with seg_off[0] = 0 and
// seg_off[S] = M (voxel s owns rows [seg_off[s], seg_off[s+1])), out [S][C].  One wave per voxel,
// lane = channel (+64 per pass), so every row is one coalesced 4*C-byte read; four rows in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

__global__ __launch_bounds__(256) void voxels_sum_fwd_kernel(const float* __restrict__ x,
                                                             const int32_t* __restrict__ seg_off, int S, int C,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int r0 = seg_off[s], r1 = seg_off[s + 1];
    for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
        int r = r0;
        for (; r + 4 <= r1; r += 4) {  // loads issued together, added in row order
            const float a0 = x[(size_t)r * C + c];
            const float a1 = x[(size_t)(r + 1) * C + c];
            const float a2 = x[(size_t)(r + 2) * C + c];
            const float a3 = x[(size_t)(r + 3) * C + c];
            acc += a0;
            acc += a1;
            acc += a2;
            acc += a3;
        }
        for (; r < r1; ++r) acc += x[(size_t)r * C + c];
        out[(size_t)s * C + c] = acc;
    }
}

__global__ __launch_bounds__(256) void voxels_sum_bwd_kernel(const float* __restrict__ grad_out,
                                                             const int32_t* __restrict__ seg_off, int S, int C,
                                                             float* __restrict__ grad_x) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int r0 = seg_off[s], r1 = seg_off[s + 1];
    for (int c = lane; c < C; c += 64) {
        const float g = grad_out[(size_t)s * C + c];
        for (int r = r0; r < r1; ++r) grad_x[(size_t)r * C + c] = g;
    }
}

}  // namespace

extern "C" {

int stp3_voxels_sum_fwd(const float* x, const int32_t* seg_off, int32_t n_segments, int32_t channels, float* out,
                        void* stream) {
    if (n_segments < 0 || channels <= 0) return STP3_EINVAL;
    if (n_segments == 0) return STP3_OK;
    if (!x || !seg_off || !out) return STP3_EINVAL;
    hipLaunchKernelGGL(voxels_sum_fwd_kernel, dim3((n_segments + 3) / 4), dim3(256), 0, (hipStream_t)stream, x,
                       seg_off, n_segments, channels, out);
    return launch_status();
}

int stp3_voxels_sum_bwd(const float* grad_out, const int32_t* seg_off, int32_t n_segments, int32_t channels,
                        float* grad_x, void* stream) {
    if (n_segments < 0 || channels <= 0) return STP3_EINVAL;
    if (n_segments == 0) return STP3_OK;
    if (!grad_out || !seg_off || !grad_x) return STP3_EINVAL;
    hipLaunchKernelGGL(voxels_sum_bwd_kernel, dim3((n_segments + 3) / 4), dim3(256), 0, (hipStream_t)stream, grad_out,
                       seg_off, n_segments, channels, grad_x);
    return launch_status();
}

}  // extern "C"
