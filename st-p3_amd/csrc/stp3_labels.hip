// stp3_labels.hip -- the BEV label side of the reference's data loader on the GPU (gfx950): SURVEY.md section 8, row f4.
//
//   stp3_fill_polygons      cv2.fillPoly of integer polygons into BEV maps -- what get_birds_eye_view_label paints the
//                           annotation boxes with (stp3/datas/NuscenesData.py:303-338: instance / segmentation / pedestrian
//                           maps) and voxelize_hd_map the road polygons (:520-564)
//   stp3_instance_labels    convert_instance_mask_to_center_and_offset_label (stp3/utils/instance.py:12-77): instance ids
//                           -> centerness heat map, offset to the instance centre, displacement to the next frame
//
// cv2 (OpenCV) is a third-party dependency that is NOT installed in this image: fillPoly is restated from its published
// algorithm (modules/imgproc/src/drawing.cpp: CollectPolyEdges + FillEdgeCollection + Line / LineIterator), PARITY
// UNPINNED -- oracle/labels_oracle.py holds the same algorithm as an edge-walking restatement (the kernel evaluates it per
// pixel in closed form), tests/test_labels_*.py check one against the other and both against Pillow's ImageDraw.polygon
// (an independent third-party rasteriser that IS installed).  What fillPoly paints, for integer vertices (shift = 0):
//   * every edge as an 8-connected Bresenham line (LineIterator, drawn left to right: err0 = M - 2 m, a diagonal step
//     whenever err < 0; M / m = the larger / smaller of |dx|, |dy|), clipped to the image;
//   * every scanline y0 <= y < y1 of the non-horizontal edges between pairs of active edges sorted by x: the pixels
//     ceil(x_left) .. floor(x_right), edge positions in 16.16 fixed point with the slope TRUNCATED to that precision.
// Polygons are painted in the order given; a later polygon overwrites an earlier one (the instance map).
//
// The instance labels are first-party arithmetic and pinned on the reference's own function (tests/golden/labels.npz).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

constexpr int kXyShift = 16;

// floor division (C++ truncates toward zero)
__device__ __forceinline__ long long floordiv(long long a, long long b) {
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

// is (x, y) a pixel of the 8-connected Bresenham line p0 -> p1 as OpenCV's LineIterator walks it (left to right)?
__device__ __forceinline__ bool on_line(int x, int y, int x0, int y0, int x1, int y1) {
    if (x0 > x1) {
        const int tx = x0, ty = y0;
        x0 = x1; y0 = y1; x1 = tx; y1 = ty;
    }
    const int dx = x1 - x0, dy = y1 - y0;
    const int sy = dy < 0 ? -1 : 1, ady = dy < 0 ? -dy : dy;
    const bool steep = ady > dx;
    const int M = steep ? ady : dx, m = steep ? dx : ady;
    const int i = steep ? (y - y0) * sy : x - x0;          // step along the major axis
    if (i < 0 || i > M) return false;
    const int c = M > 0 ? (int)(((long long)2 * m * i + M - 1) / ((long long)2 * M)) : 0;   // diagonal steps taken before step i
    return steep ? x == x0 + c : y == y0 + sy * c;
}

// one workgroup per map; the polygons of that map in paint order, every thread a share of each polygon's bounding box
__global__ __launch_bounds__(256) void fill_polygons_kernel(const stp3_poly* __restrict__ polys, int n_poly, int H, int W,
                                                            float* __restrict__ maps) {
    const int map = blockIdx.x;
    float* img = maps + (size_t)map * H * W;
    for (int p = 0; p < n_poly; ++p) {
        const stp3_poly& q = polys[p];
        if (q.map != map || q.nv < 1) continue;                      // (uniform over the workgroup)
        const int nv = q.nv;
        int xmin = q.xy[0], xmax = q.xy[0], ymin = q.xy[1], ymax = q.xy[1];
        for (int v = 1; v < nv; ++v) {
            xmin = min(xmin, q.xy[2 * v]); xmax = max(xmax, q.xy[2 * v]);
            ymin = min(ymin, q.xy[2 * v + 1]); ymax = max(ymax, q.xy[2 * v + 1]);
        }
        const int bx0 = max(xmin, 0), bx1 = min(xmax, W - 1), by0 = max(ymin, 0), by1 = min(ymax, H - 1);
        if (bx0 <= bx1 && by0 <= by1) {
            const int bw = bx1 - bx0 + 1, area = bw * (by1 - by0 + 1);
            for (int e = threadIdx.x; e < area; e += 256) {
                const int y = by0 + e / bw, x = bx0 + e % bw;
                bool paint = false;
                int n_lt = 0, n_le = 0;
                const long long X = (long long)x << kXyShift;
                for (int v = 0; v < nv; ++v) {
                    const int w = v == 0 ? nv - 1 : v - 1;
                    const int ax = q.xy[2 * w], ay = q.xy[2 * w + 1], bx = q.xy[2 * v], by = q.xy[2 * v + 1];
                    paint = paint || on_line(x, y, ax, ay, bx, by);
                    if (ay == by) continue;                          // horizontal edges take no part in the scanlines
                    // edge from its upper end (y0) down: active for y0 <= y < y1, x advances by the truncated slope
                    const int ey0 = ay < by ? ay : by, ey1 = ay < by ? by : ay;
                    if (y < ey0 || y >= ey1) continue;
                    const long long ex = (long long)(ay < by ? ax : bx) << kXyShift;
                    const long long num = ((long long)(bx - ax)) << kXyShift;
                    const long long slope = num / (by - ay);         // C++ integer division, as the reference's int64 one
                    const long long xe = ex + slope * (y - ey0);
                    n_lt += xe < X ? 1 : 0;
                    n_le += xe <= X ? 1 : 0;
                }
                // between a pair of the sorted active edges (ceil(left) <= x <= floor(right)): an odd number of crossings
                // strictly left of the pixel -- or an even number and the next edge exactly AT the pixel (its partner
                // cannot lie further left; two edges that meet at the pixel are the pair [x, x])
                paint = paint || (n_lt & 1) || (n_le != n_lt);
                if (paint) img[(size_t)y * W + x] = q.value;
            }
        }
        __syncthreads();                                             // the next polygon may overwrite these pixels
    }
}

// ---- instance labels ----------------------------------------------------------------------------------------------
// moments [2][T][K + 1][3] int32: count, sum of rows, sum of columns of every id in every frame, for the instance maps
// (first half) and for the maps warped into the previous frame (second half)
__global__ __launch_bounds__(256) void instance_moments_kernel(int T, int H, int W, int K, const int64_t* __restrict__ inst,
                                                               const float* __restrict__ warped, int32_t* __restrict__ mom) {
    const int t = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int r = p / W, c = p - r * W;
    const int64_t id = inst[(size_t)t * H * W + p];
    if (id >= 1 && id <= K) {
        int32_t* m = mom + ((size_t)t * (K + 1) + id) * 3;
        atomicAdd(m, 1); atomicAdd(m + 1, r); atomicAdd(m + 2, c);
    }
    if (warped) {
        const float wf = warped[(size_t)t * H * W + p];
        const int wid = (int)wf;
        if (wid >= 1 && wid <= K && (float)wid == wf) {
            int32_t* m = mom + (((size_t)T + t) * (K + 1) + wid) * 3;
            atomicAdd(m, 1); atomicAdd(m + 1, r); atomicAdd(m + 2, c);
        }
    }
}

__device__ __forceinline__ float mean_round(int sum, int cnt) {      // x[mask].mean().round(): float32 division, half to even
    if (sum >= (1 << 24)) return (float)rint((double)sum / (double)cnt);   // beyond float32's exact integers
    return rintf((float)sum / (float)cnt);
}

__global__ __launch_bounds__(256) void instance_labels_kernel(int T, int H, int W, int K, float ignore, float sigma2,
                                                              const int64_t* __restrict__ inst,
                                                              const int32_t* __restrict__ mom, float* __restrict__ center,
                                                              float* __restrict__ offset, float* __restrict__ flow) {
    const int t = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int r = p / W, c = p - r * W;
    const size_t plane = (size_t)H * W;
    const int32_t* mt = mom + (size_t)t * (K + 1) * 3;
    // centerness: the maximum over the instances of this frame of exp(-((xc - x)^2 + (yc - y)^2) / sigma^2)
    float best = 0.f;
    for (int id = 1; id <= K; ++id) {
        const int cnt = mt[id * 3];
        if (cnt == 0) continue;
        const float ox = mean_round(mt[id * 3 + 1], cnt) - (float)r, oy = mean_round(mt[id * 3 + 2], cnt) - (float)c;
        best = fmaxf(best, expf(-(ox * ox + oy * oy) / sigma2));
    }
    center[(size_t)t * plane + p] = best;
    const int64_t id = inst[(size_t)t * plane + p];
    float o0 = ignore, o1 = ignore, f0 = ignore, f1 = ignore;
    if (id >= 1 && id <= K) {
        const int cnt = mt[id * 3];                                   // >= 1: this pixel
        const float xc = mean_round(mt[id * 3 + 1], cnt), yc = mean_round(mt[id * 3 + 2], cnt);
        o0 = xc - (float)r;
        o1 = yc - (float)c;
        // displacement to the next frame: the instance is there too, and its mask warped into this frame is not empty
        if (t + 1 < T) {
            const int32_t* mn = mom + (size_t)(t + 1) * (K + 1) * 3 + id * 3;
            const int32_t* mw = mom + ((size_t)T + t + 1) * (K + 1) * 3 + id * 3;
            if (mn[0] > 0 && mw[0] > 0) {
                f0 = mean_round(mw[1], mw[0]) - xc;
                f1 = mean_round(mw[2], mw[0]) - yc;
            }
        }
    }
    offset[((size_t)t * 2) * plane + p] = o0;
    offset[((size_t)t * 2 + 1) * plane + p] = o1;
    flow[((size_t)t * 2) * plane + p] = f0;
    flow[((size_t)t * 2 + 1) * plane + p] = f1;
}

}  // namespace

extern "C" {

int stp3_fill_polygons(const stp3_poly* polys, int32_t n_poly, int32_t n_maps, int32_t H, int32_t W, float* maps,
                       void* stream) {
    if (n_poly < 0 || n_maps <= 0 || H <= 0 || W <= 0 || !maps || (n_poly > 0 && !polys)) return STP3_EINVAL;
    if ((int64_t)H * W >= (1LL << 31) || H > 32767 || W > 32767) return STP3_EUNSUP;     // 16.16 fixed-point edge positions
    if (n_poly == 0) return STP3_OK;
    hipLaunchKernelGGL(fill_polygons_kernel, dim3((unsigned)n_maps), dim3(256), 0, (hipStream_t)stream, polys, n_poly, H, W,
                       maps);
    return status();
}

int stp3_instance_labels_workspace_bytes(int32_t T, int32_t K, size_t* bytes) {
    if (T <= 0 || K < 0 || !bytes) return STP3_EINVAL;
    *bytes = (size_t)2 * T * (K + 1) * 3 * sizeof(int32_t);
    return STP3_OK;
}

int stp3_instance_labels(int32_t T, int32_t H, int32_t W, int32_t K, float ignore_index, float sigma,
                         const int64_t* instance, const float* warped, void* workspace, size_t workspace_bytes,
                         float* center, float* offset, float* flow, void* stream) {
    if (T <= 0 || H <= 0 || W <= 0 || K < 0 || !instance || !workspace || !center || !offset || !flow) return STP3_EINVAL;
    // an instance can cover the whole map: its coordinate sums reach H*W*max(H,W) and are accumulated in int32
    // (mean_round divides sums below 2^24 in float32 -- exact operands, the reference's arithmetic -- and larger ones,
    // where the reference's own float32 summation is no longer exact, in float64)
    if ((int64_t)H * W * (H > W ? H : W) >= (1LL << 31)) return STP3_EUNSUP;
    const size_t need = (size_t)2 * T * (K + 1) * 3 * sizeof(int32_t);
    if (workspace_bytes < need) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, need, s);
    if (e != hipSuccess) return -(int)e;
    const dim3 grid((unsigned)(((int64_t)H * W + 255) / 256), (unsigned)T);
    hipLaunchKernelGGL(instance_moments_kernel, grid, dim3(256), 0, s, T, H, W, K, instance, warped, (int32_t*)workspace);
    hipLaunchKernelGGL(instance_labels_kernel, grid, dim3(256), 0, s, T, H, W, K, ignore_index, sigma * sigma, instance,
                       (const int32_t*)workspace, center, offset, flow);
    return status();
}

}  // extern "C"
