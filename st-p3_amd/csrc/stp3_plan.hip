// stp3_plan.hip -- the planner's trajectory-cost evaluation for gfx950.
//
// Replaces, behind stp3_traj_cost_fwd / _bwd, the reference's Cost_Function (stp3/cost.py:10-47) and its seven terms
// (SafetyCost :210-241, HeadwayCost :244-272, LR_divider :274-315, Comfort :318-372, Progress :374-392, Rule :183-207,
// Cost_Volume :166-181), which the reference evaluates with ~150 small tensor operators and, for every term that looks
// at the ego footprint, a materialised (B, N, T, K) gather (K = 32 cells of the ego box, 192 of the inflated one).
// Here one thread owns one (sample, trajectory, time step): it walks the footprint tables once, reading the three BEV
// maps (occupancy, drivable area, lane dividers) and the cost volume where they lie -- an L2-resident gather workload:
// the maps of one sample are 4 x 160 KB.  Nothing is materialised; the trajectory-level terms (comfort, progress) are
// evaluated by the thread of the first time step.
//
// Arithmetic: float32, one operation per reference operation in the reference's order (no contraction: the library is
// built with -ffp-contract=off, divisions and square roots correctly rounded), truncating float -> integer conversions and
// clamps as the reference's `.long()` / `torch.clamp`.  Lane dividers: the reference takes the minimum distance to ALL
// divider cells and then drops everything beyond L = 1 m; only cells within ceil(L / resolution) of the trajectory
// point can matter, so the kernel looks at that window -- the same value, not an approximation.
//
// Backward: the trajectories, maps and target are data; the only differentiable input is the cost volume (the decoder's
// cost-volume head).  Its gradient is a scatter of the (B, N, T) cost gradients into the cells the forward read; cells
// hit by several trajectories are summed in ascending trajectory order by the thread of the FIRST of them -- no
// floating-point atomics, bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

struct PlanDims {
    int B, N, T, H, W, K0, KL;
    float dx0, dx1, bx0, bx1;
    float safety, headway, lrdivider, comfort, progress, volume, rule;
    float w0, w1, headway_dist, lr_dist;
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }   // NaN -> lo

// torch: x.long() then clamp(0, n - 1).  Values beyond the int range (and NaN, which .long() sends to INT64_MIN) end up
// at the same clamp bound either way.
__device__ __forceinline__ int cell_of(float v, int n) {
    if (!(v > -1.0f)) return 0;                            // trunc(v) <= 0, or NaN
    if (v >= (float)n) return n - 1;
    return (int)v;                                         // truncation toward zero
}

// sum over the footprint of  map_a[cell] (* map_b[cell] when map_b != nullptr; == 0 when `negate`)
template <bool kProduct, bool kNegate>
__device__ __forceinline__ float footprint_sum(const PlanDims& d, const float* __restrict__ a, const float* __restrict__ b,
                                               const int2* __restrict__ rc, int K, float fy, float fx) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const int2 o = rc[k];
        const int r = cell_of(fy + (float)o.x, d.H), c = cell_of(fx + (float)o.y, d.W);
        float v = a[r * d.W + c];
        if (kNegate) v = v == 0.f ? 1.f : 0.f;
        if (kProduct) v = v * b[r * d.W + c];
        s += v;
    }
    return s;
}

__global__ __launch_bounds__(256) void traj_cost_kernel(PlanDims d, const float* __restrict__ trajs,
                                                        const float* __restrict__ cost_volume,
                                                        const float* __restrict__ occupancy,
                                                        const float* __restrict__ drivable, const float* __restrict__ lane,
                                                        const float* __restrict__ target, const float* __restrict__ target_sum,
                                                        const int2* __restrict__ rc0, const int2* __restrict__ rcl,
                                                        float* __restrict__ cost_fc, float* __restrict__ cost_fo,
                                                        int* __restrict__ cv_cell, float* __restrict__ cv_scale) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= d.B * d.N * d.T) return;
    const int t = idx % d.T, bn = idx / d.T, b = bn / d.N;
    const float* tr = trajs + (size_t)bn * d.T * 2;
    // the reference flips the lateral axis first (cost.py:35)
    const float x = tr[2 * t] * -1.0f, y = tr[2 * t + 1];
    const float xp = t ? tr[2 * t - 2] * -1.0f : 0.f, yp = t ? tr[2 * t - 1] : 0.f;
    const float ddx = t ? x - xp : x, ddy = t ? y - yp : y;
    const float vel = sqrtf(ddx * ddx + ddy * ddy) / 0.5f;
    const size_t plane = (size_t)d.H * d.W;
    const float* occ = occupancy + ((size_t)b * d.T + t) * plane;
    const float* drv = drivable + (size_t)b * plane;
    const float fy = y / d.dx1, fx = x / d.dx0;            // get_points: trajs / dx, then rows <- y, columns <- x

    // safety (cost.py:210-241): occupied cells under the box + occupied cells under the inflated box x speed
    const float a0 = footprint_sum<false, false>(d, occ, nullptr, rc0, d.K0, fy, fx);
    const float al = footprint_sum<false, false>(d, occ, nullptr, rcl, d.KL, fy, fx);
    const float safety = clampf((a0 * d.w0 + (al * vel) * d.w1) * d.safety, 0.f, 100.f);
    // headway (:244-272): occupied AND drivable cells under the box moved L metres ahead
    const float fyh = (y + d.headway_dist) / d.dx1;
    const float headway = clampf(footprint_sum<true, false>(d, occ, drv, rc0, d.K0, fyh, fx) * d.headway, 0.f, 100.f);
    // rule (:183-207): non-drivable cells under the box
    const float rule = clampf(footprint_sum<false, true>(d, drv, nullptr, rc0, d.K0, fy, fx) * d.rule, 0.f, 100.f);
    // the trajectory point's own cell (discretize, :131-147)
    const int yi = cell_of((y - d.bx0) / d.dx0, d.H), xi = cell_of((x - d.bx1) / d.dx1, d.W);
    // lane dividers (:274-315)
    float lr = 0.f;
    {
        const float* ln = lane + (size_t)b * plane;
        const float res = fminf(d.dx0, d.dx1);
        const int rad = (int)ceilf(d.lr_dist / res);
        float best = INFINITY;
        for (int r = max(yi - rad, 0); r <= min(yi + rad, d.H - 1); ++r)
            for (int c = max(xi - rad, 0); c <= min(xi + rad, d.W - 1); ++c)
                if (ln[r * d.W + c] != 0.f) {
                    const float ey = (float)(yi - r) * d.dx1, ex = (float)(xi - c) * d.dx0;   // reversed(dx)
                    best = fminf(best, sqrtf(ey * ey + ex * ex));
                }
        if (!(best > d.lr_dist)) {
            const float g = d.lr_dist - best;
            lr = g * g;
        }
        lr = clampf(lr * d.lrdivider, 0.f, 100.f);
    }
    // cost volume (:166-181)
    const int cell = yi * d.W + xi;
    const float cv = cost_volume[((size_t)b * d.T + t) * plane + cell];
    const float term = clampf(cv, 0.f, 1000.f) * d.volume;
    const float volume = clampf(term, 0.f, 100.f);
    cost_fo[idx] = (((safety + headway) + lr) + volume) + rule;
    if (cv_cell) {
        cv_cell[idx] = cell;
        // torch.clamp passes the gradient where min <= x <= max
        cv_scale[idx] = (cv >= 0.f && cv <= 1000.f && term >= 0.f && term <= 100.f) ? d.volume : 0.f;
    }
    if (t) return;

    // ---- trajectory-level terms, by the thread of the first step: comfort (:318-372) + progress (:374-392)
    float lat_acc = 0.f, lon_acc = 0.f, jerk = 0.f, ymax = -INFINITY;
    float px = 0.f, py = 0.f, plat = 0.f, plon = 0.f, pvel = 0.f, pacc = 0.f, lx = 0.f, ly = 0.f;
    for (int i = 0; i < d.T; ++i) {
        const float cx = tr[2 * i] * -1.0f, cy = tr[2 * i + 1];
        const float sx = i ? cx - px : cx, sy = i ? cy - py : cy;
        const float lat = sx / 0.5f, lon = sy / 0.5f;
        const float v = sqrtf(sx * sx + sy * sy) / 0.5f;
        float acc = 0.f;
        if (i >= 1) {
            lat_acc = fmaxf(lat_acc, fabsf((lat - plat) / 0.5f));
            lon_acc = fmaxf(lon_acc, fabsf((lon - plon) / 0.5f));
            acc = (v - pvel) / 0.5f;
        }
        if (i >= 2) jerk = fmaxf(jerk, fabsf((acc - pacc) / 0.5f));
        ymax = fmaxf(ymax, cy);
        px = cx; py = cy; plat = lat; plon = lon; pvel = v; pacc = acc; lx = cx; ly = cy;
    }
    float comfort = 0.f;
    { const float a = clampf(lat_acc - 3.f, 0.f, 30.f); comfort += a * a; }
    { const float a = clampf(lon_acc - 3.f, 0.f, 30.f); comfort += a * a; }
    { const float a = clampf(jerk - 1.f, 0.f, 20.f); comfort += a * a; }
    comfort = clampf(comfort * d.comfort, 0.f, 100.f);
    float goal = 0.f;
    if (!(target_sum[0] < 0.5f)) {                          // the reference tests the sum over the WHOLE batch (:386)
        const float ex = lx - target[2 * b], ey = ly - target[2 * b + 1];
        goal = ex * ex + ey * ey;
    }
    const float progress = clampf((goal - ymax) * d.progress, -100.f, 100.f);
    cost_fc[bn] = comfort + progress;
}

// d cost_volume[b, t, cell] = sum over the trajectories n that read the cell of g[b, n, t] * scale[b, n, t], in ascending n.
// The (t, b) plane's N (cell, weighted gradient) pairs sit in LDS; one THREAD per trajectory scans the list (all lanes read
// the same entry: an LDS broadcast), and the first reader of a cell adds up all of them and writes the cell.  The plane
// was zero-filled by the launcher; gridDim.z workgroups share the trajectories of a plane.
__global__ __launch_bounds__(256) void traj_cost_bwd_kernel(PlanDims d, const float* __restrict__ g_fo,
                                                            const int* __restrict__ cv_cell,
                                                            const float* __restrict__ cv_scale,
                                                            float* __restrict__ d_cost_volume) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int* cells = reinterpret_cast<int*>(smem);
    float* g = reinterpret_cast<float*>(smem) + d.N;
    const int t = blockIdx.x, b = blockIdx.y;
    float* out = d_cost_volume + ((size_t)b * d.T + t) * d.H * d.W;
    for (int n = threadIdx.x; n < d.N; n += 256) {
        const size_t i = ((size_t)b * d.N + n) * d.T + t;
        cells[n] = cv_cell[i];
        g[n] = g_fo[i] * cv_scale[i];
    }
    __syncthreads();
    const int n = blockIdx.z * 256 + threadIdx.x;
    if (n >= d.N) return;
    const int c = cells[n];
    for (int m = 0; m < n; ++m)
        if (cells[m] == c) return;                         // an earlier trajectory owns this cell
    float s = g[n];
    for (int m = n + 1; m < d.N; ++m)
        if (cells[m] == c) s += g[m];
    out[c] = s;
}

bool valid(const stp3_plan_dims* p) {
    return p && p->B > 0 && p->N > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->K0 >= 0 && p->KL >= 0 && p->dx0 > 0.f &&
           p->dx1 > 0.f && p->lr_dist >= 0.f && (int64_t)p->B * p->N * p->T < (1LL << 31) &&
           (int64_t)p->B * p->T * p->H * p->W < (1LL << 31);
}

PlanDims convert(const stp3_plan_dims* p) {
    PlanDims d;
    d.B = p->B; d.N = p->N; d.T = p->T; d.H = p->H; d.W = p->W; d.K0 = p->K0; d.KL = p->KL;
    d.dx0 = p->dx0; d.dx1 = p->dx1; d.bx0 = p->bx0; d.bx1 = p->bx1;
    d.safety = p->safety; d.headway = p->headway; d.lrdivider = p->lrdivider; d.comfort = p->comfort;
    d.progress = p->progress; d.volume = p->volume; d.rule = p->rule;
    d.w0 = p->w0; d.w1 = p->w1; d.headway_dist = p->headway_dist; d.lr_dist = p->lr_dist;
    return d;
}

int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

}  // namespace

extern "C" {

int stp3_traj_cost_fwd(const stp3_plan_dims* p, const float* trajs, const float* cost_volume, const float* occupancy,
                       const float* drivable, const float* lane, const float* target, const float* target_sum,
                       const int32_t* footprint0, const int32_t* footprint_lambda, float* cost_fc, float* cost_fo,
                       int32_t* cv_cell, float* cv_scale, void* stream) {
    if (!valid(p) || !trajs || !cost_volume || !occupancy || !drivable || !lane || !target || !target_sum || !cost_fc ||
        !cost_fo || (p->K0 && !footprint0) || (p->KL && !footprint_lambda) || (!cv_cell != !cv_scale))
        return STP3_EINVAL;
    const int total = p->B * p->N * p->T;
    hipLaunchKernelGGL(traj_cost_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, convert(p), trajs,
                       cost_volume, occupancy, drivable, lane, target, target_sum, (const int2*)footprint0,
                       (const int2*)footprint_lambda, cost_fc, cost_fo, cv_cell, cv_scale);
    return status();
}

int stp3_traj_cost_bwd(const stp3_plan_dims* p, const float* grad_cost_fo, const int32_t* cv_cell, const float* cv_scale,
                       float* grad_cost_volume, void* stream) {
    if (!valid(p) || !grad_cost_fo || !cv_cell || !cv_scale || !grad_cost_volume) return STP3_EINVAL;
    const size_t lds = (size_t)p->N * 8;
    if (lds > 160 * 1024 || p->B > 65535 || p->N > 65535 * 256) return STP3_EUNSUP;   // 20 480 trajectories per sample
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_cost_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    e = hipMemsetAsync(grad_cost_volume, 0, (size_t)p->B * p->T * p->H * p->W * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(traj_cost_bwd_kernel, dim3(p->T, p->B, (p->N + 255) / 256), dim3(256), lds, (hipStream_t)stream,
                       convert(p), grad_cost_fo, cv_cell, cv_scale, grad_cost_volume);
    return status();
}

}  // extern "C"
