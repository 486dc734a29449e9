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
// Pooling plan (geometry only): per-voxel lists of column runs
// ------------------------------------------------------------------------------------------
// Along an image column (fixed camera n, feature column w, depth bin d) consecutive rows h project to the
// same BEV cell most of the time (10-20 points per run with nuScenes-like rigs).  A RUN is
// a maximal set of consecutive h with one voxel id >= 0.  The plan lists, per voxel, the runs that fall into
// it -- the pooled value of the voxel is then a PULL: sum over its runs of sum_h prob[h][d] * feat[h][:].
//   vox_off [BT][V+1]   exclusive scan of runs per voxel
//   desc    [BT][P]     uint2 per run (build scratch, arrival order): x = col << 20 | d << 14 | h0 << 7 | (len - 1)
//                                      (col = n*fW + w), y = voxel id
//   runs    [BT][P]     uint4 per run (ordered): x, y as above, z = first feature row (n*fH + h0)*fW + w,
//                                      w = first probability (col*D + d)*fH + h0 -- both relative to the frame
//                       a voxel's runs are in the contiguous range [vox_off[v], vox_off[v+1]) ... of its GROUP: the
//                       runs of a group (below) are ordered longest first, ties by x -- one canonical summation order
//   gidx    [B][V+1]    exclusive scan of the group-start flags; gidx[b][V] = number of voxel groups of sample b
//   groups  [B][V+1]    first voxel of every group (ascending), closed by V.  A GROUP is the unit of work of the
//                       forward kernel: at most 16 consecutive voxels with at most ~kGroupRuns runs over all T
//                       frames together (voxels next to the cameras hold hundreds of points, far cells none:
//                       equal-sized voxel ranges would leave a few waves with 30x the average work)
// Replaces the reference's boolean mask + argsort + cumsum differencing (stp3.py:247-257, geometry.py:302-318).
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

constexpr int kGroupVox = 16;    // voxels per group at most (one accumulator row each in the forward kernel)
constexpr int kGroupRuns = 32;   // target runs per group, all frames together

inline size_t plan_bytes(const Dims& dm) {
    return align256((size_t)dm.BT * (dm.V + 1) * 4) + align256((size_t)dm.BT * dm.P * 8) +
           2 * align256((size_t)dm.B * (dm.V + 1) * 4) + align256((size_t)dm.BT * dm.P * 16);
}

struct PlanView {
    int32_t* vox_off;
    uint2* desc;      // build scratch: runs in arrival order
    int32_t* gidx;
    int32_t* groups;
    uint4* runs;      // what the forward kernel reads: ordered, with the addresses worked out
};

inline PlanView plan_view(const Dims& dm, void* base) {
    PlanView pv;
    char* p = (char*)base;
    pv.vox_off = (int32_t*)p;
    p += align256((size_t)dm.BT * (dm.V + 1) * 4);
    pv.desc = (uint2*)p;
    p += align256((size_t)dm.BT * dm.P * 8);
    pv.gidx = (int32_t*)p;
    p += align256((size_t)dm.B * (dm.V + 1) * 4);
    pv.groups = (int32_t*)p;
    p += align256((size_t)dm.B * (dm.V + 1) * 4);
    pv.runs = (uint4*)p;
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

// (1) one thread per (bt, n, w, d) walks its image column: voxel ids in COLUMN-MAJOR order
//         vox_cm[bt][col = n*fW + w][d][h]
//     (the fH ids of a thread are contiguous; the pooling kernels read a column's probabilities / ids for a fixed
//     depth bin along h -- in pixel-major order every such value sits in its own cache line), and the number of runs
//     per voxel
__global__ __launch_bounds__(64) void plan_index_count_kernel(Dims dm, GeomArgs g, int32_t* __restrict__ vox_cm,
                                                              int32_t* __restrict__ vox_cnt) {
    extern __shared__ __attribute__((aligned(16))) int32_t ids_s[];   // [64][fH + 1]
    const int bt = blockIdx.y;
    const int q0 = blockIdx.x * 64, q = q0 + threadIdx.x;
    const int ld = dm.fH + 1;
    if (q < dm.NQ) {
        const int d = q % dm.D, col = q / dm.D;
        const int w = col % dm.fW, n = col / dm.fW;
        const int b = bt / dm.T, t = bt - b * dm.T;
        const float dep = g.ds[d];
        int prev = -1;
        for (int h = 0; h < dm.fH; ++h) {
            const int v = point_voxel(dm, g, bt, b, t, n, h, w, dep);
            ids_s[threadIdx.x * ld + h] = v;
            if (v != prev && v >= 0) atomicAdd(vox_cnt + (size_t)bt * dm.V + v, 1);
            prev = v;
        }
    }
    __syncthreads();
    // the 64 x fH ids of this workgroup are one contiguous piece of vox_cm
    const int nq = min(64, dm.NQ - q0);
    int32_t* out = vox_cm + ((size_t)bt * dm.NQ + q0) * dm.fH;
    for (int i = threadIdx.x; i < nq * dm.fH; i += 64) out[i] = ids_s[(i / dm.fH) * ld + (i % dm.fH)];
}

// (2) exclusive scan of the run counts of one frame: block j owns voxels [1024 j, 1024 j + 1024); it first adds up
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

// (3) the same walk as (1) over the stored ids: every finished run takes the next free slot of its voxel
//     (counting the voxel's counter back down to zero, so the scratch is clean for the next build)
__global__ __launch_bounds__(64) void plan_fill_kernel(Dims dm, const int32_t* __restrict__ vox_cm,
                                                       const int32_t* __restrict__ vox_off,
                                                       int32_t* __restrict__ vox_cnt, uint2* __restrict__ desc) {
    extern __shared__ __attribute__((aligned(16))) int32_t ids_s[];   // [64][fH + 1]
    const int bt = blockIdx.y;
    const int q0 = blockIdx.x * 64, q = q0 + threadIdx.x;
    const int ld = dm.fH + 1;
    const int nq = min(64, dm.NQ - q0);
    const int32_t* in = vox_cm + ((size_t)bt * dm.NQ + q0) * dm.fH;
    for (int i = threadIdx.x; i < nq * dm.fH; i += 64) ids_s[(i / dm.fH) * ld + (i % dm.fH)] = in[i];
    __syncthreads();
    if (q >= dm.NQ) return;
    const int d = q % dm.D, col = q / dm.D;
    const int32_t* v0 = ids_s + threadIdx.x * ld;
    int prev = -1, h0 = 0;
    for (int h = 0; h <= dm.fH; ++h) {
        const int v = h < dm.fH ? v0[h] : -1;
        if (v != prev) {
            if (prev >= 0) {
                const int pos = atomicAdd(vox_cnt + (size_t)bt * dm.V + prev, -1) - 1;
                const int slot = vox_off[(size_t)bt * (dm.V + 1) + prev] + pos;
                desc[(size_t)bt * dm.P + slot] =
                    make_uint2(((unsigned)col << 20) | ((unsigned)d << 14) | ((unsigned)h0 << 7) | (unsigned)(h - h0 - 1),
                               (unsigned)prev);
            }
            h0 = h;
            prev = v;
        }
    }
}

// (4) work-balanced voxel groups.  The work in front of voxel v of sample b is the sum over its frames of vox_off[v]
//     (runs of all earlier voxels); a group starts where the 16-voxel block or the kGroupRuns-sized work bucket
//     changes.  flags -> exclusive scan (plan_scan_kernel) -> list of group starts.
__global__ __launch_bounds__(256) void plan_group_flags_kernel(Dims dm, const int32_t* __restrict__ vox_off,
                                                               int32_t* __restrict__ flags) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= dm.V) return;
    int before = 0, before_prev = 0;
    for (int t = 0; t < dm.T; ++t) {
        const int32_t* off = vox_off + (size_t)(b * dm.T + t) * (dm.V + 1);
        before += off[v];
        before_prev += v > 0 ? off[v - 1] : 0;
    }
    const bool start = v == 0 || (v / kGroupVox) != ((v - 1) / kGroupVox) ||
                       (before / kGroupRuns) != (before_prev / kGroupRuns);
    flags[(size_t)b * dm.V + v] = start ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_group_list_kernel(Dims dm, const int32_t* __restrict__ flags,
                                                              const int32_t* __restrict__ gidx,
                                                              int32_t* __restrict__ groups) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v > dm.V) return;
    const int32_t* gi = gidx + (size_t)b * (dm.V + 1);
    int32_t* gl = groups + (size_t)b * (dm.V + 1);
    if (v == dm.V) gl[gi[dm.V]] = dm.V;                      // closes the last group
    else if (flags[(size_t)b * dm.V + v]) gl[gi[v]] = v;
}

// (5) one WAVE per (group, frame) orders the group's runs (a contiguous range of `desc`; a few dozen entries, up to a
//     few hundred for a voxel next to a camera): longest first, ties by (camera, column, depth bin, first row).  The
//     atomics in (3) hand out the slots in arrival order; this makes the summation order canonical (bit-reproducible
//     pooling), and it puts runs of similar length next to each other: the forward kernel sums four consecutive runs
//     in lockstep.  <= 64 runs: rank by counting smaller keys (registers); <= kSortCap: bitonic network in LDS;
//     beyond (not seen): one lane, insertion sort.
constexpr int kSortCap = 2048;

__device__ __forceinline__ unsigned long long sort_word(uint2 e) {   // ascending word = longest first, then col|d|h0
    const unsigned key = ((127u - (e.x & 127u)) << 25) | (e.x >> 7);
    return ((unsigned long long)key << 32) | e.y;
}
__device__ __forceinline__ uint2 unsort_word(unsigned long long wd) {
    const unsigned key = (unsigned)(wd >> 32);
    return make_uint2(((key & 0x1ffffffu) << 7) | (127u - (key >> 25)), (unsigned)wd);
}

__global__ __launch_bounds__(128) void plan_group_sort_kernel(Dims dm, const int32_t* __restrict__ vox_off,
                                                              const int32_t* __restrict__ gidx,
                                                              const int32_t* __restrict__ groups,
                                                              const uint2* __restrict__ desc,
                                                              uint4* __restrict__ runs) {
    __shared__ unsigned long long sbuf[2][kSortCap];
    const float inv_fw = 1.0f / (float)dm.fW;
    auto full = [&](uint2 e) {                                   // the run with its two addresses worked out
        const int h0 = (e.x >> 7) & 127u, d = (e.x >> 14) & 63u, col = e.x >> 20;
        const int n = (int)(((float)col + 0.5f) * inv_fw);       // col / fW (exact: col, fW < 4096)
        const int w = col - n * dm.fW;
        return make_uint4(e.x, e.y, (unsigned)((n * dm.fH + h0) * dm.fW + w), (unsigned)((col * dm.D + d) * dm.fH + h0));
    };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.z;
    const int ngroups = gidx[(size_t)b * (dm.V + 1) + dm.V];
    const int32_t* gl = groups + (size_t)b * (dm.V + 1);
    const int bt = b * dm.T + t;
    const int32_t* off = vox_off + (size_t)bt * (dm.V + 1);
    unsigned long long* sb = sbuf[wv];
    for (int g = blockIdx.x * 2 + wv; g < ngroups; g += gridDim.x * 2) {
        const int start = off[gl[g]], n = off[gl[g + 1]] - start;      // wave-uniform
        if (n < 1) continue;
        const uint2* seg = desc + (size_t)bt * dm.P + start;
        uint4* dst = runs + (size_t)bt * dm.P + start;
        if (n <= 64) {
            const unsigned long long e = lane < n ? sort_word(seg[lane]) : ~0ull;
            const unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                             (unsigned)__builtin_amdgcn_readlane((int)lo, j);
                rank += o < e ? 1 : 0;
            }
            if (lane < n) dst[rank] = full(unsort_word(e));            // the words are distinct: ranks are a permutation
        } else if (n <= kSortCap) {
            int m = 128;
            while (m < n) m <<= 1;
            for (int i = lane; i < m; i += 64) sb[i] = i < n ? sort_word(seg[i]) : ~0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int k = 2; k <= m; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = lane; i < m; i += 64) {
                        const int pp = i ^ j;
                        if (pp > i) {
                            const unsigned long long a = sb[i], c = sb[pp];
                            const bool up = (i & k) == 0;
                            if ((a > c) == up) { sb[i] = c; sb[pp] = a; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            for (int i = lane; i < n; i += 64) dst[i] = full(unsort_word(sb[i]));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (lane == 0) {
            for (int i = 0; i < n; ++i) {                            // insertion sort straight into `runs`
                const uint2 e = seg[i];
                const unsigned long long ke = sort_word(e);
                int j = i - 1;
                while (j >= 0 && sort_word(make_uint2(dst[j].x, dst[j].y)) > ke) {
                    dst[j + 1] = dst[j];
                    --j;
                }
                dst[j + 1] = full(e);
            }
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
// K2+K4+K5: forward -- one pass, pull form
// ------------------------------------------------------------------------------------------
// A wave owns one voxel group (<= 16 consecutive voxels of one sample) and walks the frames t = 0..T-1 (the discounted state
// bev_t = bev_{t-1} * discount + pool_t, stp3.py:296, lives in 16 registers per lane).  Lane = (slot, 4-channel
// chunk): the 4 slots work on 4 consecutive runs of the wave's list at a time; a slot walks the rows of its run,
//     acc[c] += prob[row][d] * feat[row][c]         (16 lanes x 16 bytes = one 256-byte feature row per load),
// and then adds the run vector to its voxel's pool row in LDS -- slot after slot, i.e. in list order, so every
// voxel is summed in ONE canonical order (bit-reproducible), without atomics.  Feature rows and probabilities are
// read through L2 (a frame's features are 2.6 MB: the workgroup -> voxel mapping keeps a sample on one pair of
// XCDs); the only HBM traffic besides the inputs is the BEV itself, written once as 256-byte voxel rows.
// Output layout: [B][T][V][C] (channels-last BEV); the reference's [B][T][C][V] is produced by a transpose pass
// when the caller asks for it.
__device__ __forceinline__ float4 ld4(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ld1(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Work distribution: the plan's voxel groups (<= 16 voxels, ~kGroupRuns runs over all frames, i.e. equal work) are
// dealt out statically to a chip-sized set of persistent waves.  What bounds this kernel is the gather traffic
// (~2 GB of 256-byte feature rows per launch), so it has to hit in L2: workgroup b runs on XCD b % 8, and with
// 8 % B == 0 XCD x serves the x % (8/B)-th CONTIGUOUS slice of sample x / (8/B)'s groups -- a compact part of the
// BEV, seen by two or three of the cameras -- and all waves walk the frames in lockstep order (t outer, groups inner),
// so that at any time an XCD works on one frame of a few cameras (~2-3 MB of features and probabilities, vs 4 MB of
// L2).  The price of the frame-major order: the discounted state of a group is re-read from the previous frame's
// output (written by the same lanes; 256-byte rows) instead of being kept in registers.
constexpr int kPullRows = 8;     // image rows in flight per slot

__global__ __launch_bounds__(256) void lift_pull_kernel(Dims dm, const float* __restrict__ feat,
                                                        const float* __restrict__ prob,
                                                        const int32_t* __restrict__ vox_off,
                                                        const uint4* __restrict__ runs,
                                                        const int32_t* __restrict__ gidx,
                                                        const int32_t* __restrict__ groups, float discount,
                                                        float* __restrict__ bev_cl) {
    __shared__ __attribute__((aligned(16))) float pool_s[4][kGroupVox][64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = lane >> 4;
    const int c4 = (lane & 15) * 4;
    const bool chan_ok = c4 < dm.C;
    float* pool = &pool_s[wv][0][0];
    const unsigned fstride = (unsigned)dm.fW * dm.C * 4u;
    // (samples, slice, rank, stride) of this wave
    int b_lo, b_hi, sub, xps, rank, stride;
    if ((8 % dm.B) == 0 && (gridDim.x & 7) == 0) {
        xps = 8 / dm.B;                                            // XCDs per sample
        const int xcd = blockIdx.x & 7;
        b_lo = xcd / xps; b_hi = b_lo + 1;
        sub = xcd % xps;
        rank = (blockIdx.x >> 3) * 4 + wv;
        stride = (gridDim.x >> 3) * 4;                             // waves per XCD
    } else {                                                       // any B: every wave strides over every sample
        xps = 1; sub = 0;
        b_lo = 0; b_hi = dm.B;
        rank = blockIdx.x * 4 + wv;
        stride = gridDim.x * 4;
    }

    for (int b = b_lo; b < b_hi; ++b) {
        const int32_t* glist = groups + (size_t)b * (dm.V + 1);
        const int ngroups = gidx[(size_t)b * (dm.V + 1) + dm.V];
        const int g_lo = (int)((int64_t)ngroups * sub / xps), g_hi = (int)((int64_t)ngroups * (sub + 1) / xps);
        const int mine = g_lo + rank < g_hi ? (g_hi - g_lo - rank + stride - 1) / stride : 0;   // groups of this wave
        const int items = mine * dm.T;                                 // (frame, group) pairs, frame-major
        // Three dependent fetches stand before the first feature row of an item: the list bounds of its voxels, the
        // descriptors of its first four runs, the rows themselves.  The first two are requested one item ahead
        // (bounds: two ahead), the descriptors of the next four runs while the current four are summed.
        auto item_group = [&](int k) { return g_lo + rank + (k % mine) * stride; };
        auto fetch_bound = [&](int k) -> int {                       // lane l <= 16: first run of voxel vfirst + l
            if (k >= items) return 0;
            const int g = item_group(k), t = k / mine;
            const int vfirst = glist[g], vend = glist[g + 1];
            return vox_off[(size_t)(b * dm.T + t) * (dm.V + 1) + min(vfirst + min(lane, kGroupVox), vend)];
        };
        auto fetch_desc = [&](int k, int r, int rend) -> uint4 {      // run r of item k (only if r < rend)
            uint4 ds = make_uint4(0u, 0u, 0u, 0u);
            if (k < items && r < rend) ds = runs[(size_t)(b * dm.T + k / mine) * dm.P + r];
            return ds;
        };
        int bound_cur = fetch_bound(0);
        int bound_nxt = fetch_bound(1);
        uint4 ds_cur = fetch_desc(0, __builtin_amdgcn_readlane(bound_cur, 0) + slot, __builtin_amdgcn_readlane(bound_cur, kGroupVox));

        for (int k = 0; k < items; ++k) {
            const int t = k / mine, g = item_group(k);
            const int bt = b * dm.T + t;
            const uint4* dlist = runs + (size_t)bt * dm.P;
            const unsigned frame_pix = (unsigned)bt * (unsigned)dm.NPIX, frame_pts = (unsigned)bt * (unsigned)dm.P;
            const float* prev_out = bev_cl + (size_t)(bt - 1) * dm.V * dm.C;   // read only for t > 0
            float* cur_out = bev_cl + (size_t)bt * dm.V * dm.C;
            const int vfirst = glist[g], vend = glist[g + 1];           // 1..16 voxels
            const int rbeg = __builtin_amdgcn_readlane(bound_cur, 0);
            const int rend = __builtin_amdgcn_readlane(bound_cur, kGroupVox);
            // requests for the items ahead
            const int bound_nn = fetch_bound(k + 2);
            const int nbeg = __builtin_amdgcn_readlane(bound_nxt, 0), nend = __builtin_amdgcn_readlane(bound_nxt, kGroupVox);
            const uint4 ds_first_nxt = fetch_desc(k + 1, nbeg + slot, nend);
            // the discounted state of the group's voxels: last frame's output rows (this lane wrote them itself)
            float4 st[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int v = vfirst + slot + 4 * q;
                st[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t > 0 && v < vend && chan_ok) st[q] = *reinterpret_cast<const float4*>(prev_out + (size_t)v * dm.C + c4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(pool + (slot + 4 * q) * 64 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            __builtin_amdgcn_wave_barrier();

            uint4 ds = ds_cur;
            for (int r0 = rbeg; r0 < rend; r0 += 4) {
                const int r = r0 + slot;
                const bool valid = r < rend;
                const uint4 ds_next = (r + 4 < rend) ? dlist[r + 4] : make_uint4(0u, 0u, 0u, 0u);   // for the next round
                const int len = valid ? (int)(ds.x & 127u) + 1 : 0;
                const unsigned foff = ((frame_pix + ds.z) * (unsigned)dm.C + (unsigned)(chan_ok ? c4 : 0)) * 4u;
                unsigned poff = (frame_pts + ds.w) * 4u;            // prob_cm[bt][col][d][h0 ...]: the run's rows are contiguous
                const int maxlen = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 16)),
                                       max(__builtin_amdgcn_readlane(len, 32), __builtin_amdgcn_readlane(len, 48)));
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const int last = len > 0 ? len - 1 : 0;
                // The memory pipeline takes one vector-load instruction per 16 cycles whatever its width, and this
                // kernel is bound by instruction issue: (a) one load brings the probabilities of 16 rows (lane j of
                // the slot: row i + j; one 64-byte piece of the column-major layout), each row's value is then
                // broadcast within the slot through the LDS crossbar; (b) feature rows are loaded unconditionally --
                // a slot whose run is shorter than the round's longest re-reads its last row (a cache hit) and
                // multiplies it by a zero probability -- which spares the exec-mask bookkeeping of predicated loads.
                float pvec = 0.f;
                for (int i = 0; i < maxlen; i += kPullRows) {
                    if ((i & 15) == 0) {
                        pvec = 0.f;
                        if (i + (lane & 15) < len) pvec = ld1(prob, poff + (unsigned)(lane & 15) * 4u);
                        poff += 64u;
                    }
                    float4 f[kPullRows];
#pragma unroll
                    for (int u = 0; u < kPullRows; ++u) f[u] = ld4(feat, foff + (unsigned)min(i + u, last) * fstride);
#pragma unroll
                    for (int u = 0; u < kPullRows; ++u) {
                        const float pr = __shfl(pvec, (lane & 48) | ((i + u) & 15));     // rows beyond the run carry 0
                        acc.x = fmaf(pr, f[u].x, acc.x);
                        acc.y = fmaf(pr, f[u].y, acc.y);
                        acc.z = fmaf(pr, f[u].z, acc.z);
                        acc.w = fmaf(pr, f[u].w, acc.w);
                    }
                }
                // add the four run vectors to their voxels' pool rows, in list order (slot 0 first)
                const int vi = valid ? (int)ds.y - vfirst : 0;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    if (slot == sl && valid && chan_ok) {
                        float4* cell = reinterpret_cast<float4*>(pool + vi * 64 + c4);
                        float4 q = *cell;
                        q.x += acc.x; q.y += acc.y; q.z += acc.z; q.w += acc.w;
                        *cell = q;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                ds = ds_next;
            }
            // bev_t = bev_{t-1} * discount + pool_t (stp3.py:296); rows of 4 voxels per store instruction
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vi = slot + 4 * q, v = vfirst + vi;
                const float4 pl = *reinterpret_cast<const float4*>(pool + vi * 64 + c4);
                float4 o;
                o.x = st[q].x * discount + pl.x;
                o.y = st[q].y * discount + pl.y;
                o.z = st[q].z * discount + pl.z;
                o.w = st[q].w * discount + pl.w;
                if (v < vend && chan_ok) *reinterpret_cast<float4*>(cur_out + (size_t)v * dm.C + c4) = o;
            }
            __builtin_amdgcn_wave_barrier();
            bound_cur = bound_nxt;
            bound_nxt = bound_nn;
            ds_cur = ds_first_nxt;
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

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char* stp3_version(void) { return "stp3hip 0.2 gfx950"; }


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

// what the pooling kernels can address with 32-bit byte offsets / the descriptor's bit fields
static int pool_limits(const Dims& dm) {
    if (dm.Z != 1 || dm.C > 64 || (dm.C & 3) || dm.D > 64) return STP3_EUNSUP;   // stp3.py:297-299 squeezes Z
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
    const dim3 qgrid((dm.NQ + 63) / 64, dm.BT);
    const size_t ids_lds = (size_t)64 * (dm.fH + 1) * sizeof(int32_t);
    hipLaunchKernelGGL(plan_index_count_kernel, qgrid, dim3(64), ids_lds, s, dm, g, vox_cm, counts);
    hipLaunchKernelGGL(plan_scan_kernel, dim3((dm.V + 1023) / 1024, dm.BT), dim3(1024), 0, s, dm.V, counts, pv.vox_off);
    hipLaunchKernelGGL(plan_fill_kernel, qgrid, dim3(64), ids_lds, s, dm, vox_cm, pv.vox_off, counts, pv.desc);
    // voxel groups of equal work: flags (in `groups`, reused below) -> scan -> list
    int32_t* flags = counts;                                    // zero again by now; restored to zero below
    hipLaunchKernelGGL(plan_group_flags_kernel, dim3((dm.V + 255) / 256, dm.B), dim3(256), 0, s, dm, pv.vox_off, flags);
    hipLaunchKernelGGL(plan_scan_kernel, dim3((dm.V + 1023) / 1024, dm.B), dim3(1024), 0, s, dm.V, flags, pv.gidx);
    hipLaunchKernelGGL(plan_group_list_kernel, dim3((dm.V + 256) / 256, dm.B), dim3(256), 0, s, dm, flags, pv.gidx, pv.groups);
    hipError_t e = hipMemsetAsync(flags, 0, (size_t)dm.B * dm.V * sizeof(int32_t), s);
    if (e != hipSuccess) return -(int)e;
    if (dm.T > 65535) return STP3_EUNSUP;
    hipLaunchKernelGGL(plan_group_sort_kernel, dim3(1024, dm.B, dm.T), dim3(128), 0, s, dm, pv.vox_off, pv.gidx, pv.groups,
                       pv.desc, pv.runs);
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

int stp3_lift_workspace_bytes(const stp3_lift_dims* dims, size_t* bytes) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = (size_t)dm.BT * dm.V * dm.C * sizeof(float);
    return STP3_OK;
}

static void launch_transpose(hipStream_t s, int batch, int rows, int cols, const float* in, float* out) {
    hipLaunchKernelGGL(transpose_kernel, dim3((rows + 63) / 64, (cols + 63) / 64, batch), dim3(256), 0, s, rows, cols, in,
                       out);
}

int stp3_lift_splat_fwd(const stp3_lift_dims* dims, const float* feat, const float* prob, const void* plan,
                        float discount, int bev_layout, void* workspace, size_t workspace_bytes, float* bev,
                        void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!feat || !prob || !plan || !bev) return STP3_EINVAL;
    if ((rc = pool_limits(dm))) return rc;
    if (bev_layout != STP3_BEV_CHANNELS_FIRST && bev_layout != STP3_BEV_CHANNELS_LAST) return STP3_EINVAL;
    const bool cf = bev_layout == STP3_BEV_CHANNELS_FIRST;
    const size_t need = cf ? (size_t)dm.BT * dm.V * dm.C * sizeof(float) : 0;
    if (cf && (!workspace || workspace_bytes < need)) return STP3_ENOSPACE;
    if (cf && dm.BT > 65535) return STP3_EUNSUP;
    PlanView pv = plan_view(dm, const_cast<void*>(plan));
    hipStream_t s = (hipStream_t)stream;
    float* out_cl = cf ? (float*)workspace : bev;
    // persistent waves: exactly the chip's worth of resident workgroups (a multiple of 8, so that every XCD gets the
    // same share), each striding over the voxel groups
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lift_pull_kernel, 256, 0) != hipSuccess || cus <= 0 || per_cu <= 0)
        return STP3_EUNSUP;
    const int64_t blocks = ((int64_t)cus * per_cu) & ~(int64_t)7 ? ((int64_t)cus * per_cu) & ~(int64_t)7 : 8;
    hipLaunchKernelGGL(lift_pull_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dm, feat, prob, pv.vox_off, pv.runs,
                       pv.gidx, pv.groups, discount, out_cl);
    if (cf) launch_transpose(s, dm.BT, dm.V, dm.C, out_cl, bev);          // [V][C] -> [C][V]: stp3.py:230-232 layout
    return launch_status();
}

int stp3_lift_splat_bwd(const stp3_lift_dims* dims, const void* grad_bev, int bev_layout, int grad_dtype,
                        const float* feat, const float* prob, const int32_t* vox_cm, float discount, void* workspace,
                        size_t workspace_bytes, float* grad_feat, float* grad_logits, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!grad_bev || !feat || !prob || !vox_cm || !grad_feat || !grad_logits || !workspace) return STP3_EINVAL;
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
                       prob, vox_cm, grad_feat, grad_logits);
    return launch_status();
}

}  // extern "C"
