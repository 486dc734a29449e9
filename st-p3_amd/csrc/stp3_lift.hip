// stp3_lift.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for ST-P3's LSS lift / voxel-pool path.
//
// Replaces (reference file:line) stp3/models/stp3.py:186-201 get_geometry, :215-221 depth
// softmax (x) feature outer product, :226-301 projection_to_birds_eye_view and
// stp3/utils/geometry.py:299-330 VoxelsSumming (forward and backward).  See include/stp3_hip.h
// for the contract of every entry point and DESIGN.md for layouts / rooflines.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off matters: the voxel-id arithmetic must round after every operation to be
// bit-identical with the reference's torch-CPU evaluation; the pooling kernels ask for FMAs
// explicitly with fmaf()).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {


struct Dims {
    int B, T, N, D, fH, fW, C, X, Y, Z;
    int BT, NPIX, P, V;
    int NCOL, NQ;  // image columns N*fW per frame; column-depth pairs NCOL*D
};

inline int check_dims(const stp3_lift_dims* d, Dims* o) {
    if (!d) return STP3_EINVAL;
    if (d->B <= 0 || d->T <= 0 || d->N <= 0 || d->D <= 0 || d->fH <= 0 || d->fW <= 0 || d->C <= 0 || d->X <= 0 ||
        d->Y <= 0 || d->Z <= 0)
        return STP3_EINVAL;
    o->B = d->B; o->T = d->T; o->N = d->N; o->D = d->D; o->fH = d->fH; o->fW = d->fW;
    o->C = d->C; o->X = d->X; o->Y = d->Y; o->Z = d->Z;
    o->BT = d->B * d->T;
    int64_t npix = (int64_t)d->N * d->fH * d->fW;
    int64_t p = npix * d->D;
    int64_t v = (int64_t)d->X * d->Y * d->Z;
    if (p >= (1LL << 31) || v >= (1LL << 31)) return STP3_EUNSUP;
    if (npix * d->C >= (1LL << 31)) return STP3_EUNSUP;
    o->NPIX = (int)npix; o->P = (int)p; o->V = (int)v;
    o->NCOL = d->N * d->fW; o->NQ = o->NCOL * d->D;
    return STP3_OK;
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

// ------------------------------------------------------------------------------------------
// K1/K3: frustum point -> voxel id (bit-exact), optional per-voxel histogram
// ------------------------------------------------------------------------------------------
// q_i = ((m_i0*x + m_i1*y) + m_i2*z) + t_i, float32, one rounding per operation (no FMA): this
// is what the reference's torch-CPU small-matrix bmm + in-place add evaluate to (stp3.py:197-198,
// :275-276).  The pragma (and -ffp-contract=off on the command line) forbid contraction.
__device__ __forceinline__ void affine3(const float* __restrict__ m, const float* __restrict__ t, float& x, float& y,
                                        float& z) {
#pragma clang fp contract(off)
    float q0 = m[0] * x;
    q0 = q0 + m[1] * y;
    q0 = q0 + m[2] * z;
    q0 = q0 + t[0];
    float q1 = m[3] * x;
    q1 = q1 + m[4] * y;
    q1 = q1 + m[5] * z;
    q1 = q1 + t[1];
    float q2 = m[6] * x;
    q2 = q2 + m[7] * y;
    q2 = q2 + m[8] * z;
    q2 = q2 + t[2];
    x = q0; y = q1; z = q2;
}

template <int ORDER>
__global__ __launch_bounds__(256) void voxel_index_kernel(Dims dm, const float* __restrict__ cam_m,
                                                          const float* __restrict__ cam_t,
                                                          const float* __restrict__ ego_r,
                                                          const float* __restrict__ ego_t, const float* __restrict__ xs,
                                                          const float* __restrict__ ys, const float* __restrict__ ds,
                                                          const float* __restrict__ bev_off,
                                                          const float* __restrict__ bev_res, int32_t* __restrict__ vox,
                                                          int32_t* __restrict__ counts) {
#pragma clang fp contract(off)
    const int bt = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= dm.P) return;
    const int b = bt / dm.T, t = bt - b * dm.T;
    int n, d, h, w;
    if (ORDER == STP3_VOX_REFERENCE) {
        w = idx % dm.fW;
        int r = idx / dm.fW;
        h = r % dm.fH;
        r /= dm.fH;
        d = r % dm.D;
        n = r / dm.D;
    } else {
        d = idx % dm.D;
        int pix = idx / dm.D;
        w = pix % dm.fW;
        int r = pix / dm.fW;
        h = r % dm.fH;
        n = r / dm.fH;
    }
    const float dep = ds[d];
    float x = xs[w] * dep;  // stp3.py:195
    float y = ys[h] * dep;
    float z = dep;
    const int cam = bt * dm.N + n;
    affine3(cam_m + cam * 9, cam_t + cam * 3, x, y, z);  // stp3.py:197-198
    for (int k = t; k < dm.T - 1; ++k) {                   // stp3.py:270-277
        const int e = b * dm.T + k;
        affine3(ego_r + e * 9, ego_t + e * 3, x, y, z);
    }
    // stp3.py:287-289: ((p - (start - res/2)) / res).long() -- true division, truncation
    const float gx = (x - bev_off[0]) / bev_res[0];
    const float gy = (y - bev_off[1]) / bev_res[1];
    const float gz = (z - bev_off[2]) / bev_res[2];
    // trunc(g) in [0, dim)  <=>  -1 < g < dim ; NaN fails both (stp3.py:239-246)
    const bool keep = (gx > -1.0f) && (gx < (float)dm.X) && (gy > -1.0f) && (gy < (float)dm.Y) && (gz > -1.0f) &&
                      (gz < (float)dm.Z);
    int rank = -1;
    if (keep) {
        rank = (int)gx * (dm.Y * dm.Z) + (int)gy * dm.Z + (int)gz;  // stp3.py:251-255
        if (counts) atomicAdd(counts + (size_t)bt * dm.V + rank, 1);
    }
    vox[(size_t)bt * dm.P + idx] = rank;
}

// ------------------------------------------------------------------------------------------
// Pooling plan (geometry only): the column runs, enumerated once, listed per voxel
// ------------------------------------------------------------------------------------------
// Along an image column (fixed camera n, feature column w, depth bin d) consecutive rows h project to the
// same BEV cell most of the time (10-20 points per run with nuScenes-like rigs).  A RUN is
// a maximal set of consecutive h with one voxel id >= 0.  The forward pass is a PUSH per image column (the
// column's features are read once and serve all D depth bins) followed by a PULL per voxel:
//   pass 1  one wave per column walks the rows h = 0..fH-1 with one accumulator per depth bin; where a run ends it
//           stores the run's C-vector in the run's SLOT
//   pass 2  a voxel sums the slots of its runs (fixed order: ascending slot id) and applies the discount recurrence
// The slot of a run is its position in the enumeration  frame, column, last row, depth bin  (the order pass 1 meets
// the run ends in).  The plan holds
//   vox_off  [BT][V+1]          exclusive scan of runs per voxel, per frame
//   masks    [BT][NCOL][fH]     two 64-bit words per image-column row: bit d of .x = a run of depth bin d ENDS at this
//                               row, bit d of .y = the point (d, h) falls inside the BEV grid
//   col_cnt  [BT*NCOL]          runs per column (build scratch)
//   col_off  [BT*NCOL+1]        exclusive scan of col_cnt over ALL frames = slot of the column's first run;
//                               col_off[bt*NCOL] = first slot of frame bt, col_off[BT*NCOL] = total number of runs
//   tmp      [BT*P]             per-voxel slot lists in arrival order (build scratch)
//   vox_runs [BT*P]             per-voxel slot lists, ascending: the runs of voxel v of frame bt are
//                               vox_runs[col_off[bt*NCOL] + vox_off[bt][v] ... + vox_off[bt][v+1])
//   run_desc [BT*P]             per slot: depth bin | first row << 8 | last row << 16 of the run
//   run_vox  [BT*P]             per slot: the run's voxel
// Replaces the reference's boolean mask + argsort + cumsum differencing (stp3.py:247-257, geometry.py:302-318).
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct __attribute__((aligned(16))) Mask2 {
    unsigned long long x, y;
};

struct PlanView {
    int32_t* vox_off;
    Mask2* masks;
    int32_t* col_cnt;
    int32_t* col_off;
    int32_t* tmp;
    int32_t* vox_runs;
    uint32_t* run_desc;
    int32_t* run_vox;
};

inline size_t plan_sections(const Dims& dm, size_t* o) {
    size_t p = 0;
    o[0] = p; p += align256((size_t)dm.BT * (dm.V + 1) * 4);
    o[1] = p; p += align256((size_t)dm.BT * dm.NCOL * dm.fH * 16);
    o[2] = p; p += align256((size_t)dm.BT * dm.NCOL * 4);
    o[3] = p; p += align256(((size_t)dm.BT * dm.NCOL + 1) * 4);
    o[4] = p; p += align256((size_t)dm.BT * dm.P * 4);
    o[5] = p; p += align256((size_t)dm.BT * dm.P * 4);
    o[6] = p; p += align256((size_t)dm.BT * dm.P * 4);
    o[7] = p; p += align256((size_t)dm.BT * dm.P * 4);
    return p;
}

inline size_t plan_bytes(const Dims& dm) {
    size_t o[8];
    return plan_sections(dm, o);
}

inline PlanView plan_view(const Dims& dm, void* base) {
    size_t o[8];
    plan_sections(dm, o);
    char* p = (char*)base;
    PlanView pv;
    pv.vox_off = (int32_t*)(p + o[0]);
    pv.masks = (Mask2*)(p + o[1]);
    pv.col_cnt = (int32_t*)(p + o[2]);
    pv.col_off = (int32_t*)(p + o[3]);
    pv.tmp = (int32_t*)(p + o[4]);
    pv.vox_runs = (int32_t*)(p + o[5]);
    pv.run_desc = (uint32_t*)(p + o[6]);
    pv.run_vox = (int32_t*)(p + o[7]);
    return pv;
}

struct GeomArgs {
    const float *cam_m, *cam_t, *ego_r, *ego_t, *xs, *ys, *ds, *bev_off, *bev_res;
};

// voxel id of frustum point (bt, n, h, w, d): the arithmetic of voxel_index_kernel, statement by statement
__device__ __forceinline__ int point_voxel(const Dims& dm, const GeomArgs& g, int bt, int b, int t, int n, int h, int w,
                                           float dep) {
#pragma clang fp contract(off)
    float x = g.xs[w] * dep;  // stp3.py:195
    float y = g.ys[h] * dep;
    float z = dep;
    const int cam = bt * dm.N + n;
    affine3(g.cam_m + cam * 9, g.cam_t + cam * 3, x, y, z);  // stp3.py:197-198
    for (int k = t; k < dm.T - 1; ++k) {                       // stp3.py:270-277
        const int e = b * dm.T + k;
        affine3(g.ego_r + e * 9, g.ego_t + e * 3, x, y, z);
    }
    const float gx = (x - g.bev_off[0]) / g.bev_res[0];        // stp3.py:287-289
    const float gy = (y - g.bev_off[1]) / g.bev_res[1];
    const float gz = (z - g.bev_off[2]) / g.bev_res[2];
    const bool keep = (gx > -1.0f) && (gx < (float)dm.X) && (gy > -1.0f) && (gy < (float)dm.Y) && (gz > -1.0f) &&
                      (gz < (float)dm.Z);
    return keep ? (int)gx * (dm.Y * dm.Z) + (int)gy * dm.Z + (int)gz : -1;
}

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// (1) one WORKGROUP per image column (bt, n, w), lane = depth bin, the rows dealt to the four waves in contiguous
//     pieces: the ballots over the lanes ARE the row's mask words.  Writes the column's ids in COLUMN-MAJOR order
//     vox_cm[bt][col][d][h] (one contiguous piece per column; the general backward kernel reads them along h), the
//     masks, the column's run count, and counts the runs per voxel.
__global__ __launch_bounds__(256) void plan_columns_kernel(Dims dm, GeomArgs g, int32_t* __restrict__ vox_cm,
                                                           int32_t* __restrict__ vox_cnt,
                                                           Mask2* __restrict__ masks,
                                                           int32_t* __restrict__ col_cnt, int stage) {
    extern __shared__ __attribute__((aligned(16))) int32_t ids_s[];   // [D][fH] (stage != 0)
    __shared__ int runs_s[4];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int colg = blockIdx.x;                           // column over all frames
    const int bt = colg / dm.NCOL, col = colg - bt * dm.NCOL;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const int b = bt / dm.T, t = bt - b * dm.T;
    const int d = lane;
    const bool live = d < dm.D;
    const float dep = live ? g.ds[d] : 0.f;
    int32_t* out = vox_cm + (size_t)colg * dm.D * dm.fH;
    int32_t* ids = stage ? ids_s : out;                    // very tall columns: straight to memory
    const int per = (dm.fH + 3) >> 2;
    const int h_lo = min(per * wv, dm.fH), h_hi = min(h_lo + per, dm.fH);
    int cur = (live && h_lo < h_hi) ? point_voxel(dm, g, bt, b, t, n, h_lo, w, dep) : -1;
    int runs = 0;
    for (int h = h_lo; h < h_hi; ++h) {
        const int nxt = (live && h + 1 < dm.fH) ? point_voxel(dm, g, bt, b, t, n, h + 1, w, dep) : -1;
        const bool valid = cur >= 0;
        const bool end = valid && nxt != cur;
        const unsigned long long ends_w = __ballot(end), valid_w = __ballot(valid);
        if (end) atomicAdd(vox_cnt + (size_t)bt * dm.V + cur, 1);
        if (live) ids[d * dm.fH + h] = cur;
        if (lane == 0) masks[(size_t)colg * dm.fH + h] = Mask2{ends_w, valid_w};
        runs += __popcll(ends_w);
        cur = nxt;
    }
    if (lane == 0) runs_s[wv] = runs;
    __syncthreads();
    if (threadIdx.x == 0) col_cnt[colg] = runs_s[0] + runs_s[1] + runs_s[2] + runs_s[3];
    if (!stage) return;
    for (int i = threadIdx.x; i < dm.D * dm.fH; i += 256) out[i] = ids[i];
}

// (2) exclusive scan of n counts by ONE workgroup (the run counts of all B*T*N*fW columns: a few thousand values)
__global__ __launch_bounds__(1024) void plan_scan_all_kernel(int n, const int32_t* __restrict__ cnt,
                                                             int32_t* __restrict__ off) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int first = 0; first < n; first += 1024) {
        const int i = first + tid;
        const int mine = i < n ? cnt[i] : 0;
        int incl = mine;
        for (int s = 1; s < 64; s <<= 1) {
            const int up = __shfl_up(incl, s);
            if (lane >= s) incl += up;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int wbase = carry_s;
        for (int k = 0; k < wv; ++k) wbase += wave_tot[k];
        if (i < n) off[i] = wbase + incl - mine;
        __syncthreads();
        if (tid == 1023) carry_s = wbase + incl;
        __syncthreads();
    }
    if (tid == 0) off[n] = carry_s;
}

// (3) exclusive scan of the run counts of one frame: block j owns voxels [1024 j, 1024 j + 1024); it first adds up
//     everything before its slice (coalesced, at most 4 V bytes per block) and then scans its slice
__global__ __launch_bounds__(1024) void plan_scan_kernel(int V, const int32_t* __restrict__ cnt,
                                                         int32_t* __restrict__ off) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int bt = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int32_t* c = cnt + (size_t)bt * V;
    int32_t* o = off + (size_t)bt * (V + 1);
    const int first = blockIdx.x * 1024;
    int pre = 0;
    for (int i = tid; i < first; i += 1024) pre += c[i];
    for (int s = 32; s > 0; s >>= 1) pre += __shfl_xor(pre, s);
    if (lane == 0) wave_tot[wv] = pre;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int i = 0; i < 16; ++i) s += wave_tot[i];
        base_s = s;
    }
    __syncthreads();
    const int base = base_s;
    __syncthreads();
    const int v = first + tid;
    const int mine = v < V ? c[v] : 0;
    int incl = mine;
    for (int s = 1; s < 64; s <<= 1) {
        const int up = __shfl_up(incl, s);
        if (lane >= s) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int wbase = 0;
    for (int i = 0; i < wv; ++i) wbase += wave_tot[i];
    if (v < V) o[v] = base + wbase + incl - mine;
    if (v == V - 1) o[V] = base + wbase + incl;
}

// (4) one workgroup per column again, rows dealt to the waves as in (1): every run end takes its slot (position in
//     the enumeration) and the next free place of its voxel's list (counting the voxel's counter back down to zero: the
//     scratch is clean for the next build).  The returning atomics of a wave's rows are independent of each other (the
//     rows are unrolled: they travel together).
__global__ __launch_bounds__(256) void plan_fill_kernel(Dims dm, const int32_t* __restrict__ vox_cm,
                                                        const Mask2* __restrict__ masks,
                                                        const int32_t* __restrict__ col_off,
                                                        const int32_t* __restrict__ vox_off,
                                                        int32_t* __restrict__ vox_cnt, int32_t* __restrict__ tmp,
                                                        uint32_t* __restrict__ run_desc,
                                                        int32_t* __restrict__ run_vox) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int colg = blockIdx.x;
    const int bt = colg / dm.NCOL;
    const int frame0 = col_off[bt * dm.NCOL];
    const int32_t* ids = vox_cm + (size_t)colg * dm.D * dm.fH;
    const Mask2* mk = masks + (size_t)colg * dm.fH;
    const int per = (dm.fH + 3) >> 2;
    const int h_lo = min(per * wv, dm.fH), h_hi = min(h_lo + per, dm.fH);
    // slot of the first run end of this wave's piece, and the first row of the run bin `lane` is in at its start
    int slot0 = col_off[colg];
    for (int h = 0; h < h_lo; ++h) slot0 += __popcll(mk[h].x);
    int first = -1;
    for (int h = h_lo - 1; h >= 0; --h) {                  // walk back while the run that reaches row h_lo is open
        const Mask2 m = mk[h];
        if (!((m.y >> lane) & 1ull) || ((m.x >> lane) & 1ull)) break;
        first = h;
    }
#pragma unroll 8
    for (int h = h_lo; h < h_hi; ++h) {
        const Mask2 m = mk[h];
        const unsigned long long ends_w = m.x;
        if (first < 0 && ((m.y >> lane) & 1ull)) first = h;
        if ((ends_w >> lane) & 1ull) {
            const int v = ids[lane * dm.fH + h];
            const int slot = slot0 + __popcll(ends_w & lanes_below(lane));
            run_desc[slot] = (unsigned)lane | ((unsigned)first << 8) | ((unsigned)h << 16);
            run_vox[slot] = v;
            first = -1;
            const int pos = atomicAdd(vox_cnt + (size_t)bt * dm.V + v, -1) - 1;
            tmp[(size_t)frame0 + vox_off[(size_t)bt * (dm.V + 1) + v] + pos] = slot;
        }
        slot0 += __popcll(ends_w);
    }
}

// (5) the atomics in (4) hand out the places in arrival order; this makes every voxel's list ascending -- ONE canonical
//     summation order (bit-reproducible results, no floating-point atomics anywhere).  Lane = voxel for the usual short
//     lists (rank sort: a handful of entries); a list longer than 32 is ranked by the whole wave.
__global__ __launch_bounds__(256) void plan_sort_kernel(Dims dm, const int32_t* __restrict__ vox_off,
                                                        const int32_t* __restrict__ col_off,
                                                        const int32_t* __restrict__ tmp, int32_t* __restrict__ vox_runs) {
    const int bt = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int frame0 = col_off[bt * dm.NCOL];
    const int32_t* off = vox_off + (size_t)bt * (dm.V + 1);
    int beg = 0, n = 0;
    if (v < dm.V) {
        beg = off[v];
        n = off[v + 1] - beg;
    }
    const int32_t* src = tmp + (size_t)frame0;
    int32_t* dst = vox_runs + (size_t)frame0;
    if (n <= 32) {
        for (int i = 0; i < n; ++i) {
            const int e = src[beg + i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += src[beg + j] < e ? 1 : 0;       // slots are distinct
            dst[beg + rank] = e;
        }
    }
    unsigned long long big = __ballot(n > 32);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bb = __shfl(beg, l), nn = __shfl(n, l);
        for (int i = lane; i < nn; i += 64) {
            const int e = src[bb + i];
            int rank = 0;
            for (int j = 0; j < nn; ++j) rank += src[bb + j] < e ? 1 : 0;
            dst[bb + rank] = e;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2a: softmax over depth bins, pixel-major rows of D floats (stp3.py:215)
// ------------------------------------------------------------------------------------------
template <int LPP>  // lanes per pixel, each lane owns 4 consecutive bins
__global__ __launch_bounds__(256) void depth_softmax_kernel(Dims dm, const float* __restrict__ logits,
                                                            float* __restrict__ prob_cm) {
    // one workgroup per image column (bt, n, w): the column's [D][fH] probabilities are staged in LDS and written
    // as ONE contiguous piece of prob_cm[bt][col][d][h] (the layout the pooling kernels read)
    extern __shared__ __attribute__((aligned(16))) float tile_s[];        // [D][fH]
    constexpr int PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP;
    const int col = blockIdx.x, bt = blockIdx.y;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const int D = dm.D;
    const int e0 = sub * 4;
    for (int h0 = 0; h0 < dm.fH; h0 += PPB) {
        const int h = h0 + threadIdx.x / LPP;
        const bool live = h < dm.fH;
        float v[4];
        const float* row = logits + ((size_t)bt * dm.NPIX + (size_t)(n * dm.fH + (live ? h : 0)) * dm.fW + w) * D;
        if (live && (D & 3) == 0 && e0 < D) {
            const float4 qv = *reinterpret_cast<const float4*>(row + e0);
            v[0] = qv.x; v[1] = qv.y; v[2] = qv.z; v[3] = qv.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (live && e0 + k < D) ? row[e0 + k] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
        for (int s = LPP / 2; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
        float ex[4], sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ex[k] = (e0 + k < D) ? __expf(v[k] - mx) : 0.f;
            sum += ex[k];
        }
#pragma unroll
        for (int s = LPP / 2; s > 0; s >>= 1) sum += __shfl_xor(sum, s);
        const float inv = 1.0f / sum;
        if (live) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e0 + k < D) tile_s[(e0 + k) * dm.fH + h] = ex[k] * inv;
        }
    }
    __syncthreads();
    float* out = prob_cm + ((size_t)bt * dm.NCOL + col) * D * dm.fH;
    for (int i = threadIdx.x; i < D * dm.fH; i += 256) out[i] = tile_s[i];
}

// ------------------------------------------------------------------------------------------
// K2+K4+K5: forward -- depth softmax + lift + voxel pooling + temporal accumulation, two passes
// ------------------------------------------------------------------------------------------
// Pass 1 (lift_column_kernel): one WAVE per image column (bt, n, w), lane = channel.  The wave turns the column's
// logits into probabilities (softmax over D per pixel, stp3.py:215; written out column-major for the backward pass),
// keeps them in LDS, and then walks the rows: the row's feature value f (one coalesced 256-byte read per row -- every
// feature and every logit is read from memory exactly ONCE) is multiplied into one accumulator per depth bin,
//     acc[d] += prob[h][d] * f            (prob: LDS broadcast; 0 where the point falls outside the grid),
// and where the plan's mask says that a run of bin d ends at this row, acc[d] -- the run's C-vector -- is stored in the
// run's slot (256 bytes, coalesced) and cleared.  Slots are handed out in the order the wave meets the run ends.
// Pass 2 (lift_gather_kernel): 16 lanes per voxel (4 channels each) add up the voxel's slots in ascending order and
// carry the discounted state  bev_t = bev_{t-1} * discount + pool_t  (stp3.py:296) through the T frames in
// registers; every BEV row is written once.  No atomics, one fixed summation order: bit-reproducible.
// Output layout: [B][T][V][C] (channels-last BEV); the reference's [B][T][C][V] is produced by a transpose pass
// when the caller asks for it.
constexpr int kColRows = 16;     // rows of a column staged per round (taller columns take several rounds)
// staged probability rows are rotated by 4 floats per row (bank-conflict-free column reads, no padding: the two staging
// arrays of a workgroup are exactly 32 KiB, five workgroups per CU)
__device__ __forceinline__ int prob_col(int r, int d) { return (d + 4 * r) & 63; }

// acc[4g .. 4g+3] += prob[bins 4g .. 4g+3] * f ; the bins of group g sit in lanes (4g & 15) .. + 3 of pv[g >> 2]
template <int GRP>
__device__ __forceinline__ void fma_group(float (&acc)[64], const float (&pv)[4], float f) {
    fmac_row_bcast<(4 * GRP + 0) & 15>(acc[4 * GRP + 0], pv[GRP >> 2], f);
    fmac_row_bcast<(4 * GRP + 1) & 15>(acc[4 * GRP + 1], pv[GRP >> 2], f);
    fmac_row_bcast<(4 * GRP + 2) & 15>(acc[4 * GRP + 2], pv[GRP >> 2], f);
    fmac_row_bcast<(4 * GRP + 3) & 15>(acc[4 * GRP + 3], pv[GRP >> 2], f);
}

// the runs of bins 4g .. 4g+3 that end at this row (nib: their 4 mask bits, wave-uniform): store and clear
template <int GRP>
__device__ __forceinline__ void emit_group(float (&acc)[64], unsigned nib, int lane_c, float*& sp, int C) {
    if (nib == 0u) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((nib >> k) & 1u) {
            if (lane_c >= 0) sp[lane_c] = acc[4 * GRP + k];         // sp: wave-uniform, the run's slot
            acc[4 * GRP + k] = 0.f;
            sp += C;
        }
    }
}

template <int G, int GRP>
struct ColumnRow {
    static __device__ __forceinline__ void run(float (&acc)[64], const float (&pv)[4], float f, unsigned lo, unsigned hi,
                                               int lane_c, float*& sp, int C) {
        fma_group<GRP>(acc, pv, f);
        emit_group<GRP>(acc, ((GRP < 8 ? lo : hi) >> ((4 * GRP) & 31)) & 15u, lane_c, sp, C);
        ColumnRow<G, GRP + 1>::run(acc, pv, f, lo, hi, lane_c, sp, C);
    }
};
template <int G>
struct ColumnRow<G, G> {
    static __device__ __forceinline__ void run(float (&)[64], const float (&)[4], float, unsigned, unsigned, int, float*&,
                                               int) {}
};

template <int G>   // depth bins in groups of 4: D <= 4 G
__global__ __launch_bounds__(256, G <= 12 ? 5 : 4) void lift_column_kernel(Dims dm, const float* __restrict__ feat,
                                                          const float* __restrict__ logits,
                                                          const Mask2* __restrict__ masks,
                                                          const int32_t* __restrict__ col_off,
                                                          float* __restrict__ prob_cm, float* __restrict__ slots) {
    __shared__ __attribute__((aligned(16))) float prob_s[4][kColRows][64];
    __shared__ __attribute__((aligned(16))) float feat_s[4][kColRows][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int colg = blockIdx.x * 4 + wv;
    if (colg >= dm.BT * dm.NCOL) return;
    const int bt = colg / dm.NCOL, col = colg - bt * dm.NCOL;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const int D = dm.D, C = dm.C;
    float (*ps)[64] = prob_s[wv];
    float (*fs)[64] = feat_s[wv];
    const Mask2* mk = masks + (size_t)colg * dm.fH;
    const size_t pix0 = (size_t)bt * dm.NPIX + (size_t)n * dm.fH * dm.fW + w;      // pixel (h = 0) of the column
    const float* fcol = feat + pix0 * C;        // wave-uniform bases; the per-lane parts fit 32 bits (pool_limits)
    const float* lcol = logits + pix0 * D;
    const bool chan = lane < C;
    float* sp = slots + (size_t)__builtin_amdgcn_readfirstlane(col_off[colg]) * C;   // slot of the next run that ends
    const int lane_c = chan ? lane : -1;
    float acc[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) acc[k] = 0.f;

    const int rsub = lane >> 4;                 // staging: 16 lanes per pixel, 4 bins / 4 channels per lane
    const int e0 = (lane & 15) * 4;
    for (int h0 = 0; h0 < dm.fH; h0 += kColRows) {
        const int rows = min(kColRows, dm.fH - h0);
        // ---- features and logits of the round: global -> LDS directly (no staging registers), 4 pixels per
        //      instruction.  A lane's data lands at slot `lane` of the 1 KiB piece, so the rotation of the probability
        //      rows is applied to the SOURCE: slot s of row r receives the bins 4 ((s - r) & 15) ...
#pragma unroll
        for (int i = 0; i < kColRows / 4; ++i) {
            const int r = 4 * i + rsub;
            const unsigned rel = (unsigned)(h0 + r) * (unsigned)dm.fW;      // pixel of row r, relative to the column's first
            const int gb = (((lane & 15) - r) & 15) * 4;           // first bin of this lane's slot
            if (r < rows && e0 < C) lds_dma16(fcol + (rel * (unsigned)C + (unsigned)e0), &fs[4 * i][0]);
            if (r < rows && gb < D) lds_dma16(lcol + (rel * (unsigned)D + (unsigned)gb), &ps[4 * i][0]);
        }
        unsigned long long vrow = 0ull;
        if (lane < rows) vrow = mk[h0 + lane].y;                    // "inside the grid" bits of row `lane`
        const unsigned valid_lo = (unsigned)vrow, valid_hi = (unsigned)(vrow >> 32);
        lds_dma_wait();
        __builtin_amdgcn_wave_barrier();
        // ---- softmax over the bins of each pixel (its 16 lanes), in place
#pragma unroll 1
        for (int i = 0; i < kColRows / 4; ++i) {
            const int r = 4 * i + rsub;
            const int gb = (((lane & 15) - r) & 15) * 4;
            const bool live = r < rows && gb < D;
            float4 q = *reinterpret_cast<const float4*>(&ps[r][e0]);
            if (!live) q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            const float mx = row16_max(fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
            q.x = live ? __expf(q.x - mx) : 0.f;
            q.y = live ? __expf(q.y - mx) : 0.f;
            q.z = live ? __expf(q.z - mx) : 0.f;
            q.w = live ? __expf(q.w - mx) : 0.f;
            const float inv = 1.0f / row16_sum((q.x + q.y) + (q.z + q.w));
            q.x = live ? q.x * inv : 0.f;
            q.y = live ? q.y * inv : 0.f;
            q.z = live ? q.z * inv : 0.f;
            q.w = live ? q.w * inv : 0.f;
            *reinterpret_cast<float4*>(&ps[r][e0]) = q;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- prob_cm[bt][col][d][h0 .. h0+rows): 16 consecutive rows of 4 bins per store instruction
        if (prob_cm) {
            float* out = prob_cm + (size_t)colg * D * dm.fH + h0;
            const int r = lane & (kColRows - 1);
            for (int d = lane / kColRows; d < D; d += 64 / kColRows)
                if (r < rows) out[d * dm.fH + r] = ps[r][prob_col(r, d)];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- points outside the grid contribute nothing: zero their probabilities in the staged copy
#pragma unroll 1
        for (int i = 0; i < kColRows / 4; ++i) {
            const int r = 4 * i + rsub;
            const int gb = (((lane & 15) - r) & 15) * 4;
            const unsigned vlo = __shfl(valid_lo, r), vhi = __shfl(valid_hi, r);
            const unsigned vbits = ((gb < 32 ? vlo : vhi) >> (gb & 31)) & 15u;
            float4 q = *reinterpret_cast<const float4*>(&ps[r][e0]);
            q.x = (vbits & 1u) ? q.x : 0.f;
            q.y = (vbits & 2u) ? q.y : 0.f;
            q.z = (vbits & 4u) ? q.z : 0.f;
            q.w = (vbits & 8u) ? q.w : 0.f;
            *reinterpret_cast<float4*>(&ps[r][e0]) = q;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- walk the rows
        unsigned long long ends_next = mk[h0].x;                    // wave-uniform address: scalar loads, one row ahead
        for (int i = 0; i < rows; ++i) {
            const float f = fs[i][lane];
            const unsigned long long ends_w = ends_next;
            ends_next = mk[min(h0 + i + 1, dm.fH - 1)].x;
            const unsigned lo = (unsigned)ends_w, hi = (unsigned)(ends_w >> 32);
            // the row's probabilities: lane l holds bins 16k + (l & 15), k = 0..3; bin d reaches every lane as the
            // DPP row broadcast of lane d & 15 of register d >> 4, folded into the multiply-add (stp3_dpp.h)
            float pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = (4 * k < G) ? ps[i][prob_col(i, 16 * k + (lane & 15))] : 0.f;
            ColumnRow<G, 0>::run(acc, pv, f, lo, hi, lane_c, sp, C);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Pass 1 on the matrix cores, for columns of at most 32 rows and C == 64 (the shapes of the reference's configurations):
// the run sums of a column are ONE small matrix product
//     S[run][c] = sum_h  M[run][h] * F[h][c],      M[run][h] = prob[h][bin(run)] if first(run) <= h <= last(run) else 0,
// evaluated 32 runs at a time with v_mfma_f32_32x32x2_f32 (K = 2 rows per instruction).  One WORKGROUP per column, its
// four waves share the staged logits / features (LDS DMA) and split everything else: 8 rows of the softmax each, a
// quarter of the probability write-out each, and the 32-run tiles round-robin -- a column is ~4 tiles, so the serial
// chain of a wave is stage -> 2 softmax steps -> ONE tile (14 K-steps), a quarter of what one wave per column costs
// (that variant was latency-bound: 27 us per wave on an otherwise empty GPU).  F is the B operand (lane l: rows
// 2s + (l >> 5), channel 32 nb + (l & 31)), read from LDS per K-step; the A operand is gathered from the staged
// probabilities with the run descriptors of the plan (slot order), so the 32 x 64 result tile IS the tile's 32
// consecutive slots.  No per-row bookkeeping at all: ~4 VALU instructions per MFMA.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMmaRows = 32;

__global__ __launch_bounds__(256, 6) void lift_column_mma_kernel(Dims dm, const float* __restrict__ feat,
                                                                const float* __restrict__ logits,
                                                                const int32_t* __restrict__ col_off,
                                                                const uint32_t* __restrict__ run_desc,
                                                                float* __restrict__ prob_cm,
                                                                float* __restrict__ slots) {
    __shared__ __attribute__((aligned(16))) float ps[kMmaRows][64];       // logits, then probabilities (rotated rows)
    __shared__ __attribute__((aligned(16))) float fs[kMmaRows][64];       // features
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int colg = blockIdx.x;
    const int bt = colg / dm.NCOL, col = colg - bt * dm.NCOL;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const int D = dm.D, fH = dm.fH;
    constexpr int C = 64;
    const size_t pix0 = (size_t)bt * dm.NPIX + (size_t)n * fH * dm.fW + w;      // pixel (h = 0) of the column
    const float* fcol = feat + pix0 * C;        // wave-uniform bases; the per-lane parts fit 32 bits (pool_limits)
    const float* lcol = logits + pix0 * D;
    const int rsub = lane >> 4, e0 = (lane & 15) * 4;
    const int half = lane >> 5, l32 = lane & 31;

    // ---- rows 8 wv .. 8 wv + 7: logits and features global -> LDS directly, 4 pixels per instruction (the rotation of
    //      the probability rows is applied to the source, see prob_col); rows beyond the column are zero features
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * wv + 4 * i + rsub;
        const int gb = (((lane & 15) - r) & 15) * 4;
        const unsigned rel = (unsigned)r * (unsigned)dm.fW;
        if (r < fH && gb < D) lds_dma16(lcol + (rel * (unsigned)D + (unsigned)gb), &ps[8 * wv + 4 * i][0]);
        if (r < fH) lds_dma16(fcol + (rel * (unsigned)C + (unsigned)e0), &fs[8 * wv + 4 * i][0]);
        else *reinterpret_cast<float4*>(&fs[r][e0]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int slot0 = __builtin_amdgcn_readfirstlane(col_off[colg]);
    const int nruns = __builtin_amdgcn_readfirstlane(col_off[colg + 1]) - slot0;
    // descriptor of this wave's first tile (issued with the other loads: a load behind stores waits for them)
    unsigned ds = (32 * wv + l32 < nruns) ? run_desc[slot0 + 32 * wv + l32] : 0x00000100u;   // empty: first 1 > last 0
    lds_dma_wait();
    __builtin_amdgcn_wave_barrier();
    // ---- softmax over the bins of each pixel (its 16 lanes), in place
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * wv + 4 * i + rsub;
        const int gb = (((lane & 15) - r) & 15) * 4;
        const bool live = r < fH && gb < D;
        float4 q = *reinterpret_cast<const float4*>(&ps[r][e0]);
        if (!live) q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        const float mx = row16_max(fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
        q.x = live ? __expf(q.x - mx) : 0.f;
        q.y = live ? __expf(q.y - mx) : 0.f;
        q.z = live ? __expf(q.z - mx) : 0.f;
        q.w = live ? __expf(q.w - mx) : 0.f;
        const float inv = 1.0f / row16_sum((q.x + q.y) + (q.z + q.w));
        q.x = live ? q.x * inv : 0.f;
        q.y = live ? q.y * inv : 0.f;
        q.z = live ? q.z * inv : 0.f;
        q.w = live ? q.w * inv : 0.f;
        *reinterpret_cast<float4*>(&ps[r][e0]) = q;
    }
    __syncthreads();
    // ---- prob_cm[bt][col][d][0 .. fH): 32 consecutive rows of 2 bins per store instruction, bins dealt to the waves
    if (prob_cm) {
        float* out = prob_cm + (size_t)colg * D * fH;
        for (int d = 2 * wv + half; d < D; d += 8)
            if (l32 < fH) out[d * fH + l32] = ps[l32][prob_col(l32, d)];
    }
    if (32 * wv >= nruns) return;
    // ---- this wave's tiles: 32 runs (= 32 consecutive slots) each
    const int ksteps = (fH + 1) >> 1;
    for (int r0 = 32 * wv; r0 < nruns; r0 += 128) {
        const int bin = (int)(ds & 255u), first = (int)((ds >> 8) & 255u), last = (int)(ds >> 16);
        if (r0 + 128 < nruns) ds = (r0 + 128 + l32 < nruns) ? run_desc[slot0 + r0 + 128 + l32] : 0x00000100u;
        f32x16 acc0, acc1;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc0[k] = acc1[k] = 0.f;
#pragma unroll
        for (int s = 0; s < kMmaRows / 2; ++s) {
            if (s < ksteps) {
                const int h = 2 * s + half;
                const float pr = ps[h][prob_col(h, bin)];
                const float a = ((h >= first) & (h <= last)) ? pr : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fs[h][l32], acc0, 0, 0, 0);       // B: rows 2s + half
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fs[h][32 + l32], acc1, 0, 0, 0);
            }
        }
        // D[row = (k & 3) + 8 (k >> 2) + 4 half][col = l32]: two 128-byte pieces of two slots per store
        float* tile = slots + (size_t)(slot0 + r0) * C + (4 * half * C + l32);
        const int left = nruns - r0 - 4 * half;              // rows of this half that exist
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = (k & 3) + 8 * (k >> 2);
            if (row < left) {
                tile[row * C] = acc0[k];
                tile[row * C + 32] = acc1[k];
            }
        }
    }
}

// OUT = float: the reference's BEV type (stp3.py:230-232).  OUT = uint16_t: the same values rounded ONCE to bf16
// (nearest even) -- what the bf16 temporal model makes of the float32 tensor in its first operator anyway: the kernel
// then writes half the bytes and the consumer's cast pass (read 4, write 2 bytes per element) disappears.
template <typename OUT, int TM>    // TM: frames handled by the prefetching form (T <= TM), 0: any T, frame by frame
__global__ __launch_bounds__(256) void lift_gather_kernel(Dims dm, const float* __restrict__ slots,
                                                          const int32_t* __restrict__ vox_off,
                                                          const int32_t* __restrict__ col_off,
                                                          const int32_t* __restrict__ vox_runs, float discount,
                                                          OUT* __restrict__ bev_cl) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c4 = (threadIdx.x & 15) * 4;
    if (v >= dm.V || c4 >= dm.C) return;
    // The chain  offsets -> list entry -> slot row  is three dependent loads per frame; walked frame by frame that
    // was 3 T memory latencies per thread and the kernel sat at 3.5 TB/s with SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.92.  The
    // frames are independent up to the final recurrence, so the offsets of ALL frames are fetched first, then the first
    // list entry of every frame, then the first slot row of every frame (most voxels hold 0-2 runs per frame): three
    // latencies in total.  The summation order (slots ascending, frames ascending) is unchanged: same bits.
    float4 st = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (TM > 0) {
        constexpr int TA = TM > 0 ? TM : 1;
        int beg[TA], end[TA], first[TA];
        const int32_t* lists[TA];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            beg[t] = end[t] = 0;
            if (t < dm.T) {
                const int bt = b * dm.T + t;
                const int32_t* off = vox_off + (size_t)bt * (dm.V + 1) + v;
                beg[t] = off[0];
                end[t] = off[1];
                lists[t] = vox_runs + (size_t)col_off[bt * dm.NCOL];
            }
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) first[t] = (t < dm.T && beg[t] < end[t]) ? lists[t][beg[t]] : -1;
        float4 q0[TA];
#pragma unroll
        for (int t = 0; t < TM; ++t)
            q0[t] = first[t] >= 0 ? *reinterpret_cast<const float4*>(slots + (size_t)first[t] * dm.C + c4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (t < dm.T) {
                float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);
                if (first[t] >= 0) {
                    pool.x += q0[t].x; pool.y += q0[t].y; pool.z += q0[t].z; pool.w += q0[t].w;
                    for (int i = beg[t] + 1; i < end[t]; ++i) {
                        const float4 q = *reinterpret_cast<const float4*>(slots + (size_t)lists[t][i] * dm.C + c4);
                        pool.x += q.x; pool.y += q.y; pool.z += q.z; pool.w += q.w;
                    }
                }
                st.x = st.x * discount + pool.x;              // stp3.py:296
                st.y = st.y * discount + pool.y;
                st.z = st.z * discount + pool.z;
                st.w = st.w * discount + pool.w;
                OUT* dst = bev_cl + ((size_t)(b * dm.T + t) * dm.V + v) * dm.C + c4;
                if constexpr (sizeof(OUT) == 4) {
                    *reinterpret_cast<float4*>(dst) = st;
                } else {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16(st.x, st.y), pack_bf16(st.z, st.w));
                }
            }
        }
        return;
    }
    for (int t = 0; t < dm.T; ++t) {
        const int bt = b * dm.T + t;
        const int32_t* off = vox_off + (size_t)bt * (dm.V + 1) + v;
        const int beg = off[0], end = off[1];
        const int32_t* list = vox_runs + (size_t)col_off[bt * dm.NCOL];
        float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = beg; i < end; ++i) {
            const float4 q = *reinterpret_cast<const float4*>(slots + (size_t)list[i] * dm.C + c4);
            pool.x += q.x; pool.y += q.y; pool.z += q.z; pool.w += q.w;
        }
        st.x = st.x * discount + pool.x;              // stp3.py:296
        st.y = st.y * discount + pool.y;
        st.z = st.z * discount + pool.z;
        st.w = st.w * discount + pool.w;
        OUT* dst = bev_cl + ((size_t)bt * dm.V + v) * dm.C + c4;
        if constexpr (sizeof(OUT) == 4) {
            *reinterpret_cast<float4*>(dst) = st;
        } else {
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16(st.x, st.y), pack_bf16(st.z, st.w));
        }
    }
}

// out[bt][c][r] = in[bt][r][c]  (rows x cols per batch entry; 64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const float* __restrict__ in,
                                                        float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const size_t base = (size_t)blockIdx.z * rows * cols;
    for (int i = wv; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + lane;
        tile[i][lane] = (r < rows && c < cols) ? in[base + (size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = wv; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + lane;
        if (r < rows && c < cols) out[base + (size_t)c * rows + r] = tile[lane][i];
    }
}

// ------------------------------------------------------------------------------------------
// K6: backward
// ------------------------------------------------------------------------------------------
// (a) gradient import: G_t[v][c] = sum_{t' >= t} discount^(t'-t) dL/dout[b][t'][v][c] (the adjoint of the discounted
//     accumulation, Horner from the last frame), written voxel-major [B*T][V][C] float32 so that the gather in (b)
//     reads one 256-byte row per run.  The incoming gradient has to be converted anyway (it arrives in the layout /
//     dtype of whatever consumed the BEV): this pass takes it as it comes -- channels-last float32 / bfloat16 or the
//     reference's channels-first float32 -- and folds the recurrence into the conversion.
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

template <bool BF16>
__global__ __launch_bounds__(256) void grad_import_cl_kernel(Dims dm, const void* __restrict__ dout, float discount,
                                                             float* __restrict__ gacc) {
    // lane = 4 consecutive channels of one voxel row (16 lanes x 16 bytes = one 256-byte row per load instruction),
    // 4 voxel rows per thread, all T frames of them requested together
    const int b = blockIdx.y;
    const int cq = threadIdx.x & 15, vr = threadIdx.x >> 4;
    const int c = cq * 4;
    constexpr int kRows = 4, kFrames = 4;
    const int v0 = blockIdx.x * (16 * kRows) + vr;
    if (c >= dm.C) return;
    float4 acc[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int hi = dm.T - 1; hi >= 0; hi -= kFrames) {
        float4 x[kFrames][kRows];
#pragma unroll
        for (int u = 0; u < kFrames; ++u)
#pragma unroll
            for (int i = 0; i < kRows; ++i) {
                const int t = hi - u, v = v0 + 16 * i;
                x[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && v < dm.V) {
                    const size_t e = ((size_t)(b * dm.T + t) * dm.V + v) * dm.C + c;
                    if (BF16) {
                        const uint2 q = *reinterpret_cast<const uint2*>((const uint16_t*)dout + e);
                        x[u][i] = make_float4(bf16_to_f32((uint16_t)(q.x & 0xffffu)), bf16_to_f32((uint16_t)(q.x >> 16)),
                                              bf16_to_f32((uint16_t)(q.y & 0xffffu)), bf16_to_f32((uint16_t)(q.y >> 16)));
                    } else {
                        x[u][i] = *reinterpret_cast<const float4*>((const float*)dout + e);
                    }
                }
            }
#pragma unroll
        for (int u = 0; u < kFrames; ++u)
#pragma unroll
            for (int i = 0; i < kRows; ++i) {
                const int t = hi - u, v = v0 + 16 * i;
                if (t >= 0 && v < dm.V) {
                    acc[i].x = acc[i].x * discount + x[u][i].x;
                    acc[i].y = acc[i].y * discount + x[u][i].y;
                    acc[i].z = acc[i].z * discount + x[u][i].z;
                    acc[i].w = acc[i].w * discount + x[u][i].w;
                    *reinterpret_cast<float4*>(gacc + ((size_t)(b * dm.T + t) * dm.V + v) * dm.C + c) = acc[i];
                }
            }
    }
}

constexpr int kTilePad = 65;

// channels-first input [B][T][C][V]: 64 x 64 (channel x voxel) tiles through LDS, coalesced on both sides
__global__ __launch_bounds__(256) void grad_import_cf_kernel(Dims dm, const float* __restrict__ dout, float discount,
                                                             float* __restrict__ gacc) {
    __shared__ float stage[64 * kTilePad];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int v0 = blockIdx.x * 64;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int t = dm.T - 1; t >= 0; --t) {
        const int bt = b * dm.T + t;
        const int v = v0 + lane;
        for (int ci = 0; ci < 16; ++ci) {
            const int c = wv * 16 + ci;
            float g = 0.f;
            if (c < dm.C && v < dm.V) g = dout[((size_t)bt * dm.C + c) * dm.V + v];
            stage[c * kTilePad + lane] = g;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int vl = wv * 16 + i;
            acc[i] = acc[i] * discount + stage[lane * kTilePad + vl];
            if (v0 + vl < dm.V && lane < dm.C) gacc[((size_t)bt * dm.V + v0 + vl) * dm.C + lane] = acc[i];
        }
        __syncthreads();
    }
}

// (b) gather, four lanes per image row:
//       dfeat[pix][c] = sum_d prob[pix][d] * G[vox(pix,d)][c]      dprob[pix][d] = sum_c feat[pix][c] * G[vox(pix,d)][c]
//     then the softmax backward dlogit = p * (dprob - sum_d p * dprob), all in one kernel.
// A wave takes a slice of <= 16 rows of one image column (n, w): lanes l, l+16, l+32, l+48 share pixel row l and own
// 16 channels each of its feature row and of its dfeat row (2 x 16 registers: 4-5 waves per SIMD).  Both products are
// per-lane FMA chains; the four partial dot products of a pixel meet with two cross-lane adds per depth bin (the
// round-1 kernel spent 7 DPP steps per point).  Rows of a column fall into the same voxel in runs, so a depth bin
// needs only one or two distinct gradient rows.  The kernel works in steps of kBwdBins depth bins: the run starts of
// every bin are found with one ballot, every start lane drops its voxel id into the step's run table in LDS at the
// run's number (bins in order, rows ascending), the rows of up to kBwdSlots runs (16 lanes x 16 bytes each) are
// requested together into registers WHILE the previous step is multiplied, then handed to the wave through LDS
// (ds_read_b128, broadcast within a run).  More runs than kBwdSlots in a step (rare) take extra, un-prefetched
// staging passes.
constexpr int kBwdBins = 8;      // depth bins per step
constexpr int kBwdRounds = 4;    // staging rounds per step: 4 runs (one per 16-lane slot) each
constexpr int kBwdSlots = 4 * kBwdRounds;
constexpr int kBwdLd = 68;       // floats per staged row: 272 bytes, consecutive rows start 4 banks apart
constexpr int kBwdTable = kBwdBins * 16;   // run table of a step: at most one run per (bin, row)

__global__ __launch_bounds__(256) void lift_bwd_kernel(Dims dm, int rows_pc, int chunks,
                                                       const float* __restrict__ gacc,
                                                       const float* __restrict__ feat,
                                                       const float* __restrict__ prob,
                                                       const int32_t* __restrict__ vox_cm,
                                                       float* __restrict__ grad_feat,
                                                       float* __restrict__ grad_logits) {
    __shared__ __attribute__((aligned(16))) float ghat_s[4][kBwdSlots][kBwdLd];
    __shared__ int table_s[4][2][kBwdTable];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-contiguous task order: neighbouring columns (which fetch neighbouring / the same gradient rows) share an L2
    const int64_t ntasks = (int64_t)dm.BT * dm.NCOL * chunks;
    int64_t task;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        task = (int64_t)logical * 4 + wv;
    }
    if (task >= ntasks) return;                                   // whole wave; no block-level barrier below
    const int chunk = (int)(task % chunks);
    const int col = (int)((task / chunks) % dm.NCOL);
    const int bt = (int)(task / ((int64_t)chunks * dm.NCOL));
    const int quarter = lane >> 4, hh = lane & 15;
    const int h = chunk * rows_pc + hh;
    const bool active = hh < rows_pc && h < dm.fH;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const size_t gp = (size_t)bt * dm.NPIX + (size_t)(n * dm.fH + (active ? h : 0)) * dm.fW + w;
    const unsigned le_mask = (2u << hh) - 1u;                     // rows <= mine
    const int slot = quarter, c4 = hh * 4;                        // staging role: run `slot` of a round, channels c4 .. c4+3
    const int cbase = quarter * 16;                               // compute role: channels cbase .. cbase+15
    const float* grows = gacc + (size_t)bt * dm.V * dm.C;
    float (*ghat)[kBwdLd] = ghat_s[wv];

    float4 f[4], df[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        df[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active && cbase + k * 4 < dm.C) f[k] = *reinterpret_cast<const float4*>(feat + gp * dm.C + cbase + k * 4);
    }

    // voxel ids / probabilities of the kBwdBins bins from d0 on (beyond D: v = -1, p = 0); column-major layouts
    // [bt][col][d][h]: the 16 rows of a bin are one 64-byte piece
    const size_t cm0 = ((size_t)bt * dm.NCOL + col) * dm.D * dm.fH + (active ? h : 0);
    auto load_ids = [&](int d0, int (&v)[kBwdBins]) {
#pragma unroll
        for (int j = 0; j < kBwdBins; ++j) v[j] = (active && d0 + j < dm.D) ? vox_cm[cm0 + (size_t)(d0 + j) * dm.fH] : -1;
    };
    auto load_probs = [&](int d0, float (&p)[kBwdBins]) {
#pragma unroll
        for (int j = 0; j < kBwdBins; ++j) p[j] = (active && d0 + j < dm.D) ? prob[cm0 + (size_t)(d0 + j) * dm.fH] : 0.f;
    };
    // Run starts of every bin of a step; the start lanes (quarter 0) enter their voxel id in the step's run table.
    // my[j] = number of this lane's run within the step (runs numbered bin by bin, rows ascending).  Returns the total.
    auto index_runs = [&](const int (&v)[kBwdBins], int* table, int (&my)[kBwdBins]) -> int {
        int seen = 0;
#pragma unroll
        for (int j = 0; j < kBwdBins; ++j) {
            int prev = __shfl_up(v[j], 1);
            if (hh == 0) prev = -2;                               // a slice's first row always starts a run
            const bool start = active && v[j] >= 0 && v[j] != prev;
            const unsigned sb = (unsigned)__ballot(start) & 0xffffu;     // quarter 0; the others mirror it
            my[j] = seen + __builtin_popcount(sb & le_mask) - 1;
            if (start && quarter == 0) table[my[j]] = v[j];
            seen += __builtin_popcount(sb);
        }
        return seen;
    };
    // request the rows of runs first .. first + kBwdSlots - 1 of a step (slot s of round q = run first + 4 q + s)
    auto request = [&](const int* table, int total, int first, float4 (&raw)[kBwdRounds]) {
#pragma unroll
        for (int q = 0; q < kBwdRounds; ++q) {
            const int r = first + 4 * q + slot;
            raw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < total && c4 < dm.C) raw[q] = *reinterpret_cast<const float4*>(grows + (size_t)table[r] * dm.C + c4);
        }
    };

    // lanes without a run read staged row 0 and multiply it by zero: it must hold finite numbers from the start
    if (slot == 0) *reinterpret_cast<float4*>(&ghat[0][c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    float p[kBwdBins];
    int v[kBwdBins], vn[kBwdBins];
    int my[kBwdBins], myn[kBwdBins];
    float4 raw[kBwdRounds];
    load_ids(0, v);
    int total = index_runs(v, table_s[wv][0], my);
    __builtin_amdgcn_wave_barrier();
    request(table_s[wv][0], total, 0, raw);
    float sdot = 0.f;

    for (int d0 = 0, step = 0; d0 < dm.D; d0 += kBwdBins, ++step) {
        const int* table = table_s[wv][step & 1];
        load_probs(d0, p);                                         // used after the hand-over below
        // hand the requested rows of this step to the wave
#pragma unroll
        for (int q = 0; q < kBwdRounds; ++q)
            if (4 * q + slot < total && c4 < dm.C) *reinterpret_cast<float4*>(&ghat[4 * q + slot][c4]) = raw[q];
        __builtin_amdgcn_wave_barrier();
        // index the next step's runs before this one is multiplied (its rows are requested after the multiply:
        // `raw` is free then, and the loads fly during the epilogue of this step and the prologue of the next)
        const bool more = d0 + kBwdBins < dm.D;
        int total_n = 0;
        if (more) {
            load_ids(d0 + kBwdBins, vn);
            total_n = index_runs(vn, table_s[wv][(step + 1) & 1], myn);
        }
        float t8[kBwdBins];
#pragma unroll
        for (int j = 0; j < kBwdBins; ++j) t8[j] = 0.f;

        for (int base = 0; base < (total > 0 ? total : 1); base += kBwdSlots) {
            if (base > 0) {                                        // overflow pass: stage runs base .. base+kBwdSlots-1 now
                __builtin_amdgcn_wave_barrier();
                request(table, total, base, raw);                   // `raw` is free: this step's rows were handed over
#pragma unroll
                for (int q = 0; q < kBwdRounds; ++q)
                    if (base + 4 * q + slot < total && c4 < dm.C) *reinterpret_cast<float4*>(&ghat[4 * q + slot][c4]) = raw[q];
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int j = 0; j < kBwdBins; ++j) {
                const int loc = my[j] - base;
                const bool mine = active && v[j] >= 0 && loc >= 0 && loc < kBwdSlots;
                const float* grow = &ghat[mine ? loc : 0][cbase];   // lanes without a run read row 0 and drop the result
                const float pj = mine ? p[j] : 0.f;
                float dp = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (cbase + k * 4 < dm.C) {
                        const float4 g = *reinterpret_cast<const float4*>(grow + k * 4);
                        dp = fmaf(f[k].x, g.x, dp);
                        dp = fmaf(f[k].y, g.y, dp);
                        dp = fmaf(f[k].z, g.z, dp);
                        dp = fmaf(f[k].w, g.w, dp);
                        df[k].x = fmaf(pj, g.x, df[k].x);
                        df[k].y = fmaf(pj, g.y, df[k].y);
                        df[k].z = fmaf(pj, g.z, df[k].z);
                        df[k].w = fmaf(pj, g.w, df[k].w);
                    }
                }
                t8[j] += mine ? dp : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (more) request(table_s[wv][(step + 1) & 1], total_n, 0, raw);
        // first pass of the softmax backward: p * dprob (the correction needs the whole sum over d)
#pragma unroll
        for (int j = 0; j < kBwdBins; ++j) {
            float dp = t8[j];
            dp += __shfl_xor(dp, 16);                              // the other quarters of the same pixel
            dp += __shfl_xor(dp, 32);
            t8[j] = p[j] * dp;
            sdot += t8[j];
        }
        if (active && quarter == 0) {
#pragma unroll
            for (int j = 0; j < kBwdBins; j += 4) {
                if ((dm.D & 3) == 0) {
                    if (d0 + j < dm.D)
                        *reinterpret_cast<float4*>(grad_logits + gp * dm.D + d0 + j) = make_float4(t8[j], t8[j + 1], t8[j + 2], t8[j + 3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (d0 + j + i < dm.D) grad_logits[gp * dm.D + d0 + j + i] = t8[j + i];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kBwdBins; ++j) {
            v[j] = vn[j];
            my[j] = myn[j];
        }
        total = total_n;
    }
    if (!active) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (cbase + k * 4 < dm.C) *reinterpret_cast<float4*>(grad_feat + gp * dm.C + cbase + k * 4) = df[k];
    if (quarter != 0) return;
    // dlogit = p * dprob - p * sum_d (p * dprob)
    if ((dm.D & 3) == 0) {
        for (int d0 = 0; d0 < dm.D; d0 += 4) {
            float4 tq = *reinterpret_cast<float4*>(grad_logits + gp * dm.D + d0);
            tq.x -= prob[cm0 + (size_t)(d0 + 0) * dm.fH] * sdot;
            tq.y -= prob[cm0 + (size_t)(d0 + 1) * dm.fH] * sdot;
            tq.z -= prob[cm0 + (size_t)(d0 + 2) * dm.fH] * sdot;
            tq.w -= prob[cm0 + (size_t)(d0 + 3) * dm.fH] * sdot;
            *reinterpret_cast<float4*>(grad_logits + gp * dm.D + d0) = tq;
        }
    } else {
        for (int d = 0; d < dm.D; ++d) grad_logits[gp * dm.D + d] -= prob[cm0 + (size_t)d * dm.fH] * sdot;
    }
}

// ------------------------------------------------------------------------------------------
// Backward on the matrix cores (columns of at most 32 rows, C == 64): the adjoint of lift_column_mma_kernel + gather
// ------------------------------------------------------------------------------------------
// One workgroup per image column, 32 runs (one tile) at a time.  With G[run][c] = the gradient row of the run's voxel
// (G_t = sum_{t' >= t} discount^(t'-t) dbev_{t'}: summed on the fly over the frames of a channels-last gradient, so the
// import pass and its 2 x 123 MB disappear; a channels-first gradient arrives pre-summed from grad_import_cf_kernel),
// M[run][h] = prob[h][bin(run)] on the run's rows and F the column's features:
//     dM[run][h]  = sum_c G[run][c] F[h][c]        four 16 x 16 output tiles, one per wave (v_mfma_f32_16x16x4_f32)
//     dF[h][c]   += sum_run M[run][h] G[run][c]    wave = (channel half, run half)      (v_mfma_f32_32x32x2_f32)
// dM is scattered to dP[h][bin] (every point belongs to at most one run), and after the last tile the softmax backward
// dlogit = P (dP - <P, dP>) runs per pixel.  The probabilities are recomputed from the logits exactly as in the forward
// (same code, same bits), so the forward does not have to write them.  Gradient rows are fetched once per run (the
// first backward fetched them once per run and row slice) into a double-buffered LDS tile, the next tile's rows
// travelling while the current tile is multiplied.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kGLd = 68;          // floats per staged gradient row: 16-byte aligned, rows 4 banks apart
constexpr int kBwdRuns = 256;     // run descriptors / voxels staged per pass (a column has ~110 runs)

template <bool BF16, bool SUMMED>   // gradient dtype; SUMMED: `grad` is G_t itself ([BT][V][C] float32)
__global__ __launch_bounds__(256, 3) void lift_bwd_column_kernel(Dims dm, const void* __restrict__ grad, float discount,
                                                                const float* __restrict__ feat,
                                                                const float* __restrict__ logits,
                                                                const int32_t* __restrict__ col_off,
                                                                const uint32_t* __restrict__ run_desc,
                                                                const int32_t* __restrict__ run_vox,
                                                                float* __restrict__ grad_feat,
                                                                float* __restrict__ grad_logits) {
    __shared__ __attribute__((aligned(16))) float ps[kMmaRows][64];       // logits, then probabilities (rotated rows)
    __shared__ __attribute__((aligned(16))) float fs[kMmaRows][64];       // features
    __shared__ __attribute__((aligned(16))) float dps[kMmaRows][64];      // dP, same rotation as ps
    __shared__ __attribute__((aligned(16))) float gs[2][32][kGLd];        // gradient rows of the current / next tile
    __shared__ unsigned rd_s[kBwdRuns];
    __shared__ int rv_s[kBwdRuns];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int colg = blockIdx.x;
    const int bt = colg / dm.NCOL, col = colg - bt * dm.NCOL;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const int b = bt / dm.T, t = bt - b * dm.T;
    const int D = dm.D, fH = dm.fH;
    constexpr int C = 64;
    const size_t pix0 = (size_t)bt * dm.NPIX + (size_t)n * fH * dm.fW + w;
    const float* fcol = feat + pix0 * C;
    const float* lcol = logits + pix0 * D;
    const int rsub = lane >> 4, e0 = (lane & 15) * 4;
    const int half = lane >> 5, l32 = lane & 31;
    const int l16 = lane & 15, q16 = lane >> 4;

    // ---- stage the column exactly like the forward (rows 8 wv .. 8 wv + 7 per wave)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * wv + 4 * i + rsub;
        const int gb = ((l16 - r) & 15) * 4;
        const unsigned rel = (unsigned)r * (unsigned)dm.fW;
        if (r < fH && gb < D) lds_dma16(lcol + (rel * (unsigned)D + (unsigned)gb), &ps[8 * wv + 4 * i][0]);
        if (r < fH) lds_dma16(fcol + (rel * (unsigned)C + (unsigned)gb), &fs[8 * wv + 4 * i][0]);     // rotated like ps
        else *reinterpret_cast<float4*>(&fs[r][e0]) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&dps[r][e0]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int slot0 = __builtin_amdgcn_readfirstlane(col_off[colg]);
    const int nruns = __builtin_amdgcn_readfirstlane(col_off[colg + 1]) - slot0;
    // the gradient rows of 32 runs from run `first_run` (8 per wave, 4 per instruction): G_t of the run's voxel
    const float* gf = (const float*)grad;
    const uint16_t* gh = (const uint16_t*)grad;
    struct Rows { float4 g[2]; };
    auto load_rows = [&](int first_run, int chunk_base) -> Rows {
        Rows out;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int run = first_run + 8 * wv + 4 * i + rsub;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (run < min(nruns, chunk_base + kBwdRuns)) {
                const int v = rv_s[run - chunk_base];
                if (SUMMED) {
                    g = *reinterpret_cast<const float4*>(gf + ((size_t)bt * dm.V + v) * C + e0);
                } else {
                    for (int tt = dm.T - 1; tt >= t; --tt) {         // stp3.py:296, adjoint of the recurrence
                        const size_t e = ((size_t)(b * dm.T + tt) * dm.V + v) * C + e0;
                        float4 x;
                        if (BF16) {
                            const uint2 qv = *reinterpret_cast<const uint2*>(gh + e);
                            x = make_float4(bf16_to_f32((uint16_t)(qv.x & 0xffffu)), bf16_to_f32((uint16_t)(qv.x >> 16)),
                                            bf16_to_f32((uint16_t)(qv.y & 0xffffu)), bf16_to_f32((uint16_t)(qv.y >> 16)));
                        } else {
                            x = *reinterpret_cast<const float4*>(gf + e);
                        }
                        g.x = g.x * discount + x.x;
                        g.y = g.y * discount + x.y;
                        g.z = g.z * discount + x.z;
                        g.w = g.w * discount + x.w;
                    }
                }
            }
            out.g[i] = g;
        }
        return out;
    };

    // dF accumulators of this wave: channel half nb, run half kh of every tile
    const int nb = wv & 1, kh = wv >> 1;
    f32x16 dF;
#pragma unroll
    for (int k = 0; k < 16; ++k) dF[k] = 0.f;
    const int mi = wv & 1, nj = wv >> 1;          // dM: runs 16 mi .. + 15, rows 16 nj .. + 15 of the tile

    for (int base = 0; base < nruns; base += kBwdRuns) {
        if (base > 0) __syncthreads();            // rd_s / rv_s free again
        const int cnt = min(kBwdRuns, nruns - base);
        if (tid < cnt) {
            rd_s[tid] = run_desc[slot0 + base + tid];
            rv_s[tid] = run_vox[slot0 + base + tid];
        }
        if (base == 0) lds_dma_wait();            // (waits for the two loads above as well)
        __syncthreads();
        // the gradient rows travel two tiles ahead of the tile that is being multiplied
        Rows ga = load_rows(base, base), gb2 = load_rows(base + 32, base);
        if (base == 0) {
            // ---- softmax over the bins of each pixel (its 16 lanes), in place -- while the first rows travel
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = 8 * wv + 4 * i + rsub;
                const int gb = ((l16 - r) & 15) * 4;
                const bool live = r < fH && gb < D;
                float4 q = *reinterpret_cast<const float4*>(&ps[r][e0]);
                if (!live) q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                const float mx = row16_max(fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
                q.x = live ? __expf(q.x - mx) : 0.f;
                q.y = live ? __expf(q.y - mx) : 0.f;
                q.z = live ? __expf(q.z - mx) : 0.f;
                q.w = live ? __expf(q.w - mx) : 0.f;
                const float inv = 1.0f / row16_sum((q.x + q.y) + (q.z + q.w));
                q.x = live ? q.x * inv : 0.f;
                q.y = live ? q.y * inv : 0.f;
                q.z = live ? q.z * inv : 0.f;
                q.w = live ? q.w * inv : 0.f;
                *reinterpret_cast<float4*>(&ps[r][e0]) = q;
            }
        }
        for (int t0 = 0; t0 < cnt; t0 += 32) {
            const int buf = (t0 >> 5) & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<float4*>(&gs[buf][8 * wv + 4 * i + rsub][e0]) = ga.g[i];
            ga = gb2;
            gb2 = load_rows(base + t0 + 64, base);
            __syncthreads();                      // gs[buf] (and, first time, ps / dps) complete; gs[buf ^ 1] free next time
            const float (*g)[kGLd] = gs[buf];
            // ---- dM tile (runs 16 mi + i, rows 16 nj + j), K = 64 channels
            f32x4 dm4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s)
                dm4 = __builtin_amdgcn_mfma_f32_16x16x4f32(g[16 * mi + l16][4 * s + q16],
                                                           fs[16 * nj + l16][prob_col(16 * nj + l16, 4 * s + q16)], dm4, 0, 0, 0);
            // D[i = 4 q16 + k][j = l16]: run 16 mi + 4 q16 + k, row h = 16 nj + l16
            {
                const int h = 16 * nj + l16;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rr = t0 + 16 * mi + 4 * q16 + k;
                    const unsigned ds = rr < cnt ? rd_s[rr] : 0x00000100u;
                    const int bin = (int)(ds & 255u), first = (int)((ds >> 8) & 255u), last = (int)(ds >> 16);
                    if ((h >= first) & (h <= last)) dps[h][prob_col(h, bin)] = dm4[k];
                }
            }
            // ---- dF += M^T G over the runs 16 kh .. 16 kh + 15 of the tile, channels 32 nb ..
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int rr = 16 * kh + 2 * s + half;                // run within the tile (the K index)
                const unsigned ds = t0 + rr < cnt ? rd_s[t0 + rr] : 0x00000100u;
                const int bin = (int)(ds & 255u), first = (int)((ds >> 8) & 255u), last = (int)(ds >> 16);
                const float pr = ps[l32][prob_col(l32, bin)];
                const float a = ((l32 >= first) & (l32 <= last)) ? pr : 0.f;   // M^T[h = l32][run]
                dF = __builtin_amdgcn_mfma_f32_32x32x2f32(a, g[rr][32 * nb + l32], dF, 0, 0, 0);
            }
        }
    }
    if (nruns == 0) {                             // (no tile loop ran: the staged column still has to land)
        lds_dma_wait();
    }
    __syncthreads();
    // ---- dfeat: the two run halves of a channel half are added in LDS (fixed order), rows written by the kh = 0 waves
    float (*red)[kGLd] = gs[0];                   // [32 rows][64 channels]
    if (kh == 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) red[(k & 3) + 8 * (k >> 2) + 4 * half][32 * nb + l32] = dF[k];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int h = (k & 3) + 8 * (k >> 2) + 4 * half;
            if (h < fH)
                grad_feat[(pix0 + (size_t)h * dm.fW) * C + 32 * nb + l32] = dF[k] + red[h][32 * nb + l32];
        }
    }
    // ---- softmax backward per pixel (its 16 lanes): dlogit = P (dP - <P, dP>)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * wv + 4 * i + rsub;
        const int gb = ((l16 - r) & 15) * 4;
        const float4 pq = *reinterpret_cast<const float4*>(&ps[r][e0]);
        const float4 dq = *reinterpret_cast<const float4*>(&dps[r][e0]);
        const float dot = row16_sum((pq.x * dq.x + pq.y * dq.y) + (pq.z * dq.z + pq.w * dq.w));
        if (r < fH && gb < D) {
            float4 o;
            o.x = pq.x * (dq.x - dot);
            o.y = pq.y * (dq.y - dot);
            o.z = pq.z * (dq.z - dot);
            o.w = pq.w * (dq.w - dot);
            *reinterpret_cast<float4*>(grad_logits + (pix0 + (size_t)r * dm.fW) * D + gb) = o;
        }
    }
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char* stp3_version(void) { return "stp3hip 0.4 gfx950"; }


int stp3_voxel_index(const stp3_lift_dims* dims, const float* cam_m, const float* cam_t, const float* ego_r,
                     const float* ego_t, const float* xs, const float* ys, const float* ds, const float* bev_offset,
                     const float* bev_res, int order, int32_t* vox, int32_t* counts, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!cam_m || !cam_t || !ego_r || !ego_t || !xs || !ys || !ds || !bev_offset || !bev_res || !vox)
        return STP3_EINVAL;
    if (order != STP3_VOX_REFERENCE && order != STP3_VOX_PIXELMAJOR) return STP3_EINVAL;
    dim3 grid((dm.P + 255) / 256, dm.BT);
    hipStream_t s = (hipStream_t)stream;
    if (order == STP3_VOX_REFERENCE)
        hipLaunchKernelGGL(voxel_index_kernel<STP3_VOX_REFERENCE>, grid, dim3(256), 0, s, dm, cam_m, cam_t, ego_r,
                           ego_t, xs, ys, ds, bev_offset, bev_res, vox, counts);
    else
        hipLaunchKernelGGL(voxel_index_kernel<STP3_VOX_PIXELMAJOR>, grid, dim3(256), 0, s, dm, cam_m, cam_t, ego_r,
                           ego_t, xs, ys, ds, bev_offset, bev_res, vox, counts);
    return launch_status();
}

// the shapes the matrix-core kernels cover (the backward one recomputes the probabilities from the logits)
static bool column_mma_shape(const Dims& dm) { return dm.fH <= kMmaRows && dm.C == 64; }

// what the pooling kernels can address with 32-bit byte offsets / the descriptor's bit fields
static int pool_limits(const Dims& dm) {
    if (dm.Z != 1 || dm.C > 64 || (dm.C & 3) || dm.D > 64 || (dm.D & 3)) return STP3_EUNSUP;   // stp3.py:297-299 squeezes Z
    if (dm.fH > 128 || dm.NCOL >= 4096) return STP3_EUNSUP;
    const int64_t widest = (int64_t)dm.BT * dm.NPIX * (dm.C > dm.D ? dm.C : dm.D) * 4;
    if (widest >= (1LL << 32)) return STP3_EUNSUP;
    return STP3_OK;
}

int stp3_lift_plan_bytes(const stp3_lift_dims* dims, size_t* bytes) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = plan_bytes(dm);
    return STP3_OK;
}

int stp3_lift_plan_build(const stp3_lift_dims* dims, const float* cam_m, const float* cam_t, const float* ego_r,
                         const float* ego_t, const float* xs, const float* ys, const float* ds,
                         const float* bev_offset, const float* bev_res, int32_t* vox_cm, int32_t* counts, void* plan,
                         size_t plan_size, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!cam_m || !cam_t || !ego_r || !ego_t || !xs || !ys || !ds || !bev_offset || !bev_res || !vox_cm || !counts ||
        !plan)
        return STP3_EINVAL;
    if ((rc = pool_limits(dm))) return rc;
    if (plan_size < plan_bytes(dm)) return STP3_ENOSPACE;
    if (dm.BT > 65535) return STP3_EUNSUP;
    PlanView pv = plan_view(dm, plan);
    GeomArgs g{cam_m, cam_t, ego_r, ego_t, xs, ys, ds, bev_offset, bev_res};
    hipStream_t s = (hipStream_t)stream;
    const int ncols = dm.BT * dm.NCOL;
    const dim3 cgrid(ncols);
    size_t ids_lds = (size_t)dm.D * dm.fH * sizeof(int32_t);
    const int stage = ids_lds <= 48 * 1024;
    if (!stage) ids_lds = 0;
    hipLaunchKernelGGL(plan_columns_kernel, cgrid, dim3(256), ids_lds, s, dm, g, vox_cm, counts, pv.masks, pv.col_cnt, stage);
    hipLaunchKernelGGL(plan_scan_all_kernel, dim3(1), dim3(1024), 0, s, ncols, pv.col_cnt, pv.col_off);
    hipLaunchKernelGGL(plan_scan_kernel, dim3((dm.V + 1023) / 1024, dm.BT), dim3(1024), 0, s, dm.V, counts, pv.vox_off);
    hipLaunchKernelGGL(plan_fill_kernel, cgrid, dim3(256), 0, s, dm, vox_cm, pv.masks, pv.col_off, pv.vox_off, counts, pv.tmp,
                       pv.run_desc, pv.run_vox);
    hipLaunchKernelGGL(plan_sort_kernel, dim3((dm.V + 255) / 256, dm.BT), dim3(256), 0, s, dm, pv.vox_off, pv.col_off, pv.tmp,
                       pv.vox_runs);
    return launch_status();
}

int stp3_depth_softmax(const stp3_lift_dims* dims, const float* logits, float* prob, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!logits || !prob) return STP3_EINVAL;
    if (dm.D > 128) return STP3_EUNSUP;
    if (dm.BT > 65535) return STP3_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(dm.NCOL, dm.BT);
    const size_t lds = (size_t)dm.D * dm.fH * sizeof(float);
    if (lds > 64 * 1024) return STP3_EUNSUP;
    if (dm.D <= 32)
        hipLaunchKernelGGL(depth_softmax_kernel<8>, grid, dim3(256), lds, s, dm, logits, prob);
    else if (dm.D <= 64)
        hipLaunchKernelGGL(depth_softmax_kernel<16>, grid, dim3(256), lds, s, dm, logits, prob);
    else
        hipLaunchKernelGGL(depth_softmax_kernel<32>, grid, dim3(256), lds, s, dm, logits, prob);
    return launch_status();
}

// forward: one slot (C floats) per run, at most one run per frustum point, then (channels-first output only) the
// channels-last result in front of the transpose; backward: the imported gradient [BT][V][C]
static size_t workspace_need(const Dims& dm) {
    const size_t slots = (size_t)dm.BT * dm.P * dm.C * sizeof(float);
    const size_t planes = (size_t)dm.BT * dm.V * dm.C * sizeof(float);
    return slots + planes;
}

int stp3_lift_workspace_bytes(const stp3_lift_dims* dims, size_t* bytes) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = workspace_need(dm);
    return STP3_OK;
}

static void launch_transpose(hipStream_t s, int batch, int rows, int cols, const float* in, float* out) {
    hipLaunchKernelGGL(transpose_kernel, dim3((rows + 63) / 64, (cols + 63) / 64, batch), dim3(256), 0, s, rows, cols, in,
                       out);
}

int stp3_lift_splat_fwd(const stp3_lift_dims* dims, const float* feat, const float* logits, const void* plan,
                        float discount, int bev_layout, void* workspace, size_t workspace_bytes, float* prob_cm,
                        void* bev, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!feat || !logits || !plan || !bev || !workspace) return STP3_EINVAL;
    if ((rc = pool_limits(dm))) return rc;
    if (bev_layout != STP3_BEV_CHANNELS_FIRST && bev_layout != STP3_BEV_CHANNELS_LAST &&
        bev_layout != STP3_BEV_CHANNELS_LAST_BF16)
        return STP3_EINVAL;
    const bool cf = bev_layout == STP3_BEV_CHANNELS_FIRST;
    const bool out_bf16 = bev_layout == STP3_BEV_CHANNELS_LAST_BF16;
    if (((uintptr_t)bev & (out_bf16 ? 7 : 15))) return STP3_EUNSUP;
    if (workspace_bytes < workspace_need(dm)) return STP3_ENOSPACE;
    if (dm.B > 65535) return STP3_EUNSUP;
    PlanView pv = plan_view(dm, const_cast<void*>(plan));
    hipStream_t s = (hipStream_t)stream;
    float* slots = (float*)workspace;
    float* out_cl = cf ? slots + (size_t)dm.BT * dm.P * dm.C : (float*)bev;
    const int ncols = dm.BT * dm.NCOL;
    const dim3 cgrid((ncols + 3) / 4);
    if (column_mma_shape(dm))
        hipLaunchKernelGGL(lift_column_mma_kernel, dim3(ncols), dim3(256), 0, s, dm, feat, logits, pv.col_off, pv.run_desc,
                           prob_cm, slots);
    else if (dm.D <= 32)
        hipLaunchKernelGGL(lift_column_kernel<8>, cgrid, dim3(256), 0, s, dm, feat, logits, pv.masks, pv.col_off, prob_cm, slots);
    else if (dm.D <= 48)
        hipLaunchKernelGGL(lift_column_kernel<12>, cgrid, dim3(256), 0, s, dm, feat, logits, pv.masks, pv.col_off, prob_cm, slots);
    else
        hipLaunchKernelGGL(lift_column_kernel<16>, cgrid, dim3(256), 0, s, dm, feat, logits, pv.masks, pv.col_off, prob_cm, slots);
    // the prefetching form keeps the offsets, first list entries and first slot rows of all frames in registers:
    // instantiated per frame count (T = 3: 8 waves per SIMD where the 8-frame form left 5)
    const dim3 ggrid((dm.V + 15) / 16, dm.B);
#define STP3_GATHER(OUT, TM, ptr)                                                                                     \
    hipLaunchKernelGGL((lift_gather_kernel<OUT, TM>), ggrid, dim3(256), 0, s, dm, slots, pv.vox_off, pv.col_off,      \
                       pv.vox_runs, discount, ptr)
    if (out_bf16) {
        if (dm.T <= 3) STP3_GATHER(uint16_t, 3, (uint16_t*)bev);
        else if (dm.T <= 8) STP3_GATHER(uint16_t, 8, (uint16_t*)bev);
        else STP3_GATHER(uint16_t, 0, (uint16_t*)bev);
    } else {
        if (dm.T <= 3) STP3_GATHER(float, 3, out_cl);
        else if (dm.T <= 8) STP3_GATHER(float, 8, out_cl);
        else STP3_GATHER(float, 0, out_cl);
    }
#undef STP3_GATHER
    if (cf) launch_transpose(s, dm.BT, dm.V, dm.C, out_cl, (float*)bev);  // [V][C] -> [C][V]: stp3.py:230-232 layout
    return launch_status();
}

int stp3_lift_bwd_needs_prob(const stp3_lift_dims* dims, int* needs) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!needs) return STP3_EINVAL;
    *needs = column_mma_shape(dm) ? 0 : 1;
    return STP3_OK;
}

int stp3_lift_splat_bwd(const stp3_lift_dims* dims, const void* grad_bev, int bev_layout, int grad_dtype,
                        const float* feat, const float* logits, const float* prob_cm, const int32_t* vox_cm,
                        const void* plan, float discount, void* workspace, size_t workspace_bytes, float* grad_feat,
                        float* grad_logits, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!grad_bev || !feat || !grad_feat || !grad_logits || !workspace) return STP3_EINVAL;
    if ((rc = pool_limits(dm))) return rc;
    if (bev_layout != STP3_BEV_CHANNELS_FIRST && bev_layout != STP3_BEV_CHANNELS_LAST) return STP3_EINVAL;
    if (grad_dtype != STP3_DTYPE_F32 && grad_dtype != STP3_DTYPE_BF16) return STP3_EINVAL;
    const bool cf = bev_layout == STP3_BEV_CHANNELS_FIRST;
    if (cf && grad_dtype != STP3_DTYPE_F32) return STP3_EUNSUP;
    if (workspace_bytes < (size_t)dm.BT * dm.V * dm.C * sizeof(float)) return STP3_ENOSPACE;
    if (dm.B > 65535) return STP3_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    float* gacc = (float*)workspace;
    const dim3 igrid((dm.V + 63) / 64, dm.B);
    if (column_mma_shape(dm)) {
        if (!logits || !plan) return STP3_EINVAL;
        PlanView pv = plan_view(dm, const_cast<void*>(plan));
        const dim3 grid(dm.BT * dm.NCOL);
        if (cf) {       // reference layout: transpose + recurrence in the import pass, then rows of G_t
            hipLaunchKernelGGL(grad_import_cf_kernel, igrid, dim3(256), 0, s, dm, (const float*)grad_bev, discount, gacc);
            hipLaunchKernelGGL((lift_bwd_column_kernel<false, true>), grid, dim3(256), 0, s, dm, (const void*)gacc, discount,
                               feat, logits, pv.col_off, pv.run_desc, pv.run_vox, grad_feat, grad_logits);
        } else if (grad_dtype == STP3_DTYPE_BF16) {
            hipLaunchKernelGGL((lift_bwd_column_kernel<true, false>), grid, dim3(256), 0, s, dm, grad_bev, discount, feat,
                               logits, pv.col_off, pv.run_desc, pv.run_vox, grad_feat, grad_logits);
        } else {
            hipLaunchKernelGGL((lift_bwd_column_kernel<false, false>), grid, dim3(256), 0, s, dm, grad_bev, discount, feat,
                               logits, pv.col_off, pv.run_desc, pv.run_vox, grad_feat, grad_logits);
        }
        return launch_status();
    }
    if (!prob_cm || !vox_cm) return STP3_EINVAL;
    if (cf)
        hipLaunchKernelGGL(grad_import_cf_kernel, igrid, dim3(256), 0, s, dm, (const float*)grad_bev, discount, gacc);
    else if (grad_dtype == STP3_DTYPE_BF16)
        hipLaunchKernelGGL(grad_import_cl_kernel<true>, igrid, dim3(256), 0, s, dm, grad_bev, discount, gacc);
    else
        hipLaunchKernelGGL(grad_import_cl_kernel<false>, igrid, dim3(256), 0, s, dm, grad_bev, discount, gacc);
    // one wave per slice of <= 16 rows of an image column
    const int chunks = (dm.fH + 15) / 16;
    const int rows_pc = (dm.fH + chunks - 1) / chunks;
    const int64_t tasks = (int64_t)dm.BT * dm.NCOL * chunks;
    hipLaunchKernelGGL(lift_bwd_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, s, dm, rows_pc, chunks, gacc, feat,
                       prob_cm, vox_cm, grad_feat, grad_logits);
    return launch_status();
}

}  // extern "C"
