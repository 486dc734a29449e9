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

constexpr int kWave = 64;
constexpr int kSortCap = 4096;  // longest per-voxel list that is canonically ordered

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
// Pooling plan (geometry only): column runs, their destination rows, per-voxel row ranges
// ------------------------------------------------------------------------------------------
// Along an image column (fixed camera n, feature column w, depth bin d) consecutive rows h
// project to the same BEV cell most of the time (10-20 points per run with
// nuScenes-like rigs).  A RUN is a maximal set of consecutive h with one voxel id >= 0.  The
// forward pass first reduces every run to one C-vector (stage 1, camera-side, feature rows read
// once) and then sums the few run vectors of each voxel (stage 2, BEV-side).  The plan gives
//   run_base [BT][NQ+1]  exclusive scan of runs per q = (n*fW + w)*D + d; run id = run_base[q] + j
//   vox_off  [BT][V+1]   exclusive scan of runs per voxel
//   dest     [BT][P]     run id -> row of the stage-1 buffer; a voxel's rows are the contiguous
//                        range [vox_off[v], vox_off[v+1]), ordered by run id (canonical order)
//   list     [BT][P]     row -> run id (scratch of the build, kept for inspection)
struct PlanView {
    int32_t* run_base;
    int32_t* vox_off;
    int32_t* dest;
    int32_t* list;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline size_t plan_bytes(const Dims& dm) {
    return align256((size_t)dm.BT * (dm.NQ + 1) * 4) + align256((size_t)dm.BT * (dm.V + 1) * 4) +
           2 * align256((size_t)dm.BT * dm.P * 4);
}

inline PlanView plan_view(const Dims& dm, void* base) {
    PlanView pv;
    char* p = (char*)base;
    pv.run_base = (int32_t*)p;
    p += align256((size_t)dm.BT * (dm.NQ + 1) * 4);
    pv.vox_off = (int32_t*)p;
    p += align256((size_t)dm.BT * (dm.V + 1) * 4);
    pv.dest = (int32_t*)p;
    p += align256((size_t)dm.BT * dm.P * 4);
    pv.list = (int32_t*)p;
    return pv;
}

// one thread per (bt, n, w, d): count the runs of its column, histogram them per voxel
__global__ __launch_bounds__(256) void run_count_kernel(Dims dm, const int32_t* __restrict__ vox_pm,
                                                        int32_t* __restrict__ run_cnt,
                                                        int32_t* __restrict__ vox_cnt) {
    const int bt = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= dm.NQ) return;
    const int d = q % dm.D, col = q / dm.D;
    const int w = col % dm.fW, n = col / dm.fW;
    const int32_t* v0 = vox_pm + ((size_t)bt * dm.NPIX + (size_t)n * dm.fH * dm.fW + w) * dm.D + d;
    const size_t hstride = (size_t)dm.fW * dm.D;
    int prev = -1, cnt = 0;
    for (int h = 0; h < dm.fH; ++h) {
        const int v = v0[h * hstride];
        if (v != prev) {
            if (v >= 0) {
                ++cnt;
                atomicAdd(vox_cnt + (size_t)bt * dm.V + v, 1);
            }
            prev = v;
        }
    }
    run_cnt[(size_t)bt * (dm.NQ + 1) + q] = cnt;
}

// exclusive scan of n counts per bt (in place capable: in == out allowed), out[n] = total.
// zero_in != 0 additionally zeroes `in` (it then serves as the fill cursor).
__global__ __launch_bounds__(1024) void plan_scan_kernel(int n, int in_stride, int32_t* in, int32_t* out,
                                                         int zero_in) {
    __shared__ int wave_tot[16];
    const int bt = blockIdx.x, tid = threadIdx.x;
    int32_t* cnt = in + (size_t)bt * in_stride;
    int32_t* off = out + (size_t)bt * (n + 1);
    const int per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += cnt[i];
    int incl = sum;
    const int lane = tid & 63, wv = tid >> 6;
    for (int s = 1; s < 64; s <<= 1) {
        int o = __shfl_up(incl, s);
        if (lane >= s) incl += o;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += wave_tot[i];
    int run = base + incl - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = cnt[i];
        if (zero_in) cnt[i] = 0;
        off[i] = run;   // (in == out: cnt[i] was read above)
        run += c;
    }
    if (tid == 1023) off[n] = run;
}

// one thread per (bt, n, w, d): hand every run of the column a slot in its voxel's list
__global__ __launch_bounds__(256) void run_fill_kernel(Dims dm, const int32_t* __restrict__ vox_pm,
                                                       const int32_t* __restrict__ run_base,
                                                       const int32_t* __restrict__ vox_off,
                                                       int32_t* __restrict__ cursor, int32_t* __restrict__ list) {
    const int bt = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= dm.NQ) return;
    const int d = q % dm.D, col = q / dm.D;
    const int w = col % dm.fW, n = col / dm.fW;
    const int32_t* v0 = vox_pm + ((size_t)bt * dm.NPIX + (size_t)n * dm.fH * dm.fW + w) * dm.D + d;
    const size_t hstride = (size_t)dm.fW * dm.D;
    int rid = run_base[(size_t)bt * (dm.NQ + 1) + q];
    int prev = -1;
    for (int h = 0; h < dm.fH; ++h) {
        const int v = v0[h * hstride];
        if (v != prev) {
            if (v >= 0) {
                const int slot = vox_off[(size_t)bt * (dm.V + 1) + v] + atomicAdd(cursor + (size_t)bt * dm.V + v, 1);
                list[(size_t)bt * dm.P + slot] = rid++;
            }
            prev = v;
        }
    }
}

// One wave per voxel list: order the run ids ascending, so that stage 2 adds every voxel's run
// vectors in one fixed order regardless of how the atomics in run_fill_kernel interleaved.
__global__ __launch_bounds__(128) void plan_sort_kernel(Dims dm, const int32_t* __restrict__ offsets,
                                                        int32_t* __restrict__ list) {
    __shared__ int32_t sbuf[2][kSortCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t total = (int64_t)dm.BT * dm.V;
    const int64_t nwaves = (int64_t)gridDim.x * 2;
    int32_t* s = sbuf[wv];
    for (int64_t item = (int64_t)blockIdx.x * 2 + wv; item < total; item += nwaves) {
        const int bt = (int)(item / dm.V), v = (int)(item - (int64_t)bt * dm.V);
        const int32_t* off = offsets + (size_t)bt * (dm.V + 1) + v;
        const int start = __builtin_amdgcn_readfirstlane(off[0]);
        const int n = __builtin_amdgcn_readfirstlane(off[1]) - start;
        if (n < 2 || n > kSortCap) continue;
        int32_t* seg = list + (size_t)bt * dm.P + start;
        if (n <= kWave) {
            const int e = lane < n ? seg[lane] : INT_MAX;
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += (__builtin_amdgcn_readlane(e, j) < e) ? 1 : 0;
            if (lane < n) seg[rank] = e;  // entries are distinct, so ranks are a permutation
        } else {
            int m = 128;
            while (m < n) m <<= 1;
            for (int i = lane; i < m; i += kWave) s[i] = i < n ? seg[i] : INT_MAX;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int k = 2; k <= m; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = lane; i < m; i += kWave) {
                        const int p = i ^ j;
                        if (p > i) {
                            const int a = s[i], bb = s[p];
                            const bool up = (i & k) == 0;
                            if ((a > bb) == up) { s[i] = bb; s[p] = a; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            for (int i = lane; i < n; i += kWave) seg[i] = s[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// dest[list[row]] = row
__global__ __launch_bounds__(256) void plan_invert_kernel(Dims dm, const int32_t* __restrict__ vox_off,
                                                          const int32_t* __restrict__ list,
                                                          int32_t* __restrict__ dest) {
    const int bt = blockIdx.y;
    const int total = vox_off[(size_t)bt * (dm.V + 1) + dm.V];
    for (int row = blockIdx.x * 256 + threadIdx.x; row < total; row += gridDim.x * 256)
        dest[(size_t)bt * dm.P + list[(size_t)bt * dm.P + row]] = row;
}

// ------------------------------------------------------------------------------------------
// K2a: softmax over depth bins, pixel-major rows of D floats (stp3.py:215)
// ------------------------------------------------------------------------------------------
template <int LPP>  // lanes per pixel, each lane owns 4 consecutive bins
__global__ __launch_bounds__(256) void depth_softmax_kernel(int64_t npix_total, int D,
                                                            const float* __restrict__ logits,
                                                            float* __restrict__ prob) {
    constexpr int PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP;
    const int64_t pix = (int64_t)blockIdx.x * PPB + threadIdx.x / LPP;
    const bool live = pix < npix_total;
    const int e0 = sub * 4;
    float v[4];
    const float* row = logits + pix * D;
    if (live && (D & 3) == 0 && e0 < D) {
        const float4 q = *reinterpret_cast<const float4*>(row + e0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
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
    float* orow = prob + pix * D;
    if (live && (D & 3) == 0 && e0 < D) {
        *reinterpret_cast<float4*>(orow + e0) = make_float4(ex[0] * inv, ex[1] * inv, ex[2] * inv, ex[3] * inv);
    } else if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e0 + k < D) orow[e0 + k] = ex[k] * inv;
    }
}

// v_readlane_b32 on a float (the builtin is typed int: pass the bits, not the value)
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ------------------------------------------------------------------------------------------
// K2+K4: stage 1 -- depth (x) feature outer product reduced along image columns (camera side)
// ------------------------------------------------------------------------------------------
// Workgroup = one image column (bt, n, w); wave g owns the 8 depth bins [8g, 8g+8).  The column's fH
// feature rows, depth probabilities and voxel ids are staged in LDS once per workgroup.
// Lane = (bin, 8-channel chunk): a lane keeps the running sum of ITS bin for ITS 8 channels,
//   acc[k] += prob[h][bin] * feat[h][8*chunk + k],
// so one wave instruction advances 8 frustum points x 64 channels (the earlier lane = channel version
// spent 6 scalar/vector instructions per point and was issue-bound at 83 us).  When the voxel id of a bin
// changes from one image row to the next, the 8 lanes of that bin store the finished run vector as one
// 256-B row of `runs`, at the row the plan assigned to it (dest[run id]) -- ~12x fewer rows than frustum
// points, every input read once, no atomics.
template <bool VEC8>  // VEC8: C is a multiple of 8 (two float4 stores per lane)
__global__ __launch_bounds__(1024) void lift_runs_kernel(Dims dm, int Dp, const float* __restrict__ feat,
                                                         const float* __restrict__ prob,
                                                         const int32_t* __restrict__ vox_pm,
                                                         const int32_t* __restrict__ run_base,
                                                         const int32_t* __restrict__ dest,
                                                         float* __restrict__ runs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* fcol = smem;                                   // [fH][64]
    float* pcol = smem + (size_t)dm.fH * 64;              // [fH][Dp]
    int* vcol = reinterpret_cast<int*>(pcol + (size_t)dm.fH * Dp);   // [fH][Dp]
    int* dcol = vcol + (size_t)dm.fH * Dp;                // destination rows of the column's runs (<= fH*D)
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int col = blockIdx.x, bt = blockIdx.y;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const size_t pix0 = (size_t)bt * dm.NPIX + (size_t)n * dm.fH * dm.fW + w;   // pixel of row h = pix0 + h*fW

    // ---- stage the column: feature rows (zero-padded to 64 channels), probabilities, voxel ids ----
    for (int h = g; h < dm.fH; h += nwaves)
        fcol[h * 64 + lane] = lane < dm.C ? feat[(pix0 + (size_t)h * dm.fW) * dm.C + lane] : 0.f;
    for (int i = threadIdx.x; i < dm.fH * Dp; i += blockDim.x) {
        const int h = i / Dp, d = i - h * Dp;
        const size_t src = (pix0 + (size_t)h * dm.fW) * dm.D + d;
        pcol[i] = d < dm.D ? prob[src] : 0.f;
        vcol[i] = d < dm.D ? vox_pm[src] : -1;
    }
    const int32_t* dst = dest + (size_t)bt * dm.P;
    const int32_t* rb_col = run_base + (size_t)bt * (dm.NQ + 1) + (size_t)col * dm.D;
    const int col_base = rb_col[0], col_runs = rb_col[dm.D] - rb_col[0];      // the column's runs are contiguous ids
    for (int i = threadIdx.x; i < col_runs; i += blockDim.x) dcol[i] = dst[col_base + i];
    __syncthreads();

    const int bin = lane >> 3, chunk = lane & 7;
    const int d = g * 8 + bin;
    const bool bin_ok = d < dm.D;
    float* out = runs + (size_t)bt * dm.P * dm.C + chunk * 8;
    const int* drow = dcol + (bin_ok ? rb_col[d] - col_base : 0);   // rows of this lane's bin, in run order
    int cnt = 0;                                              // runs flushed so far
    int cur = -1;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;

    auto flush = [&]() {
        const int row = drow[cnt];
        float* o = out + (size_t)row * dm.C;
        if (VEC8) {
            if (chunk * 8 < dm.C) {
                *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (chunk * 8 + k < dm.C) o[k] = acc[k];
        }
    };

    const float4* f4 = reinterpret_cast<const float4*>(fcol) + chunk * 2;
    const float* pc = pcol + d;
    const int* vc = vcol + d;
    for (int h = 0; h < dm.fH; ++h) {
        const float p = pc[h * Dp];
        const int v = vc[h * Dp];
        const float4 a = f4[h * 16], b = f4[h * 16 + 1];
        const bool changed = v != cur;                         // run boundary of this lane's bin
        if (changed && cur >= 0) {
            flush();
            ++cnt;
        }
        cur = v;
        // branch-free restart of the running sum (the store above only READS acc)
        acc[0] = fmaf(p, a.x, changed ? 0.f : acc[0]); acc[1] = fmaf(p, a.y, changed ? 0.f : acc[1]);
        acc[2] = fmaf(p, a.z, changed ? 0.f : acc[2]); acc[3] = fmaf(p, a.w, changed ? 0.f : acc[3]);
        acc[4] = fmaf(p, b.x, changed ? 0.f : acc[4]); acc[5] = fmaf(p, b.y, changed ? 0.f : acc[5]);
        acc[6] = fmaf(p, b.z, changed ? 0.f : acc[6]); acc[7] = fmaf(p, b.w, changed ? 0.f : acc[7]);
    }
    if (cur >= 0) flush();
}

// ------------------------------------------------------------------------------------------
// K4+K5: stage 2 -- per-voxel sum of run vectors, discounted accumulation over t, BEV planes
// ------------------------------------------------------------------------------------------
// A workgroup owns 64 consecutive voxels of one sample; wave w owns voxels [16w, 16w+16), whose
// run rows are ONE contiguous range of `runs` (the plan sorted the rows by voxel), streamed with
// up to 8 independent 256-B row loads in flight and added in row order (canonical => bit
// reproducible).  The running bev*discount + pool_t lives in an LDS tile [voxel][channel] that is
// written out transposed, i.e. as 256-B coalesced rows of the reference's [C][X*Y] planes.
constexpr int kTileV = 64;
constexpr int kTilePad = 65;

__global__ __launch_bounds__(256) void lift_gather_kernel(Dims dm, const float* __restrict__ runs,
                                                          const int32_t* __restrict__ vox_off, float discount,
                                                          float* __restrict__ bev) {
    __shared__ float tile[kTileV * kTilePad];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int v0 = blockIdx.x * kTileV;
    const bool chan = lane < dm.C;

    for (int i = threadIdx.x; i < kTileV * kTilePad; i += 256) tile[i] = 0.f;
    __syncthreads();

    for (int t = 0; t < dm.T; ++t) {
        const int bt = b * dm.T + t;
        const float* rbt = runs + (size_t)bt * dm.P * dm.C + lane;
        const int32_t* obt = vox_off + (size_t)bt * (dm.V + 1);
        // lane i <= 16 holds the first row of voxel v0 + 16*wv + i (clamped to the last offset)
        const int vfirst = v0 + wv * 16;
        const int bound = obt[min(vfirst + min(lane, 16), dm.V)];
        const int rbeg = __builtin_amdgcn_readlane(bound, 0);
        const int rend = __builtin_amdgcn_readlane(bound, 16);
        int vi = 0;                                              // voxel (0..15) the open sum belongs to
        int next = __builtin_amdgcn_readlane(bound, 1);          // first row of voxel vi + 1
        float acc = 0.f;
        float* cells = tile + (wv * 16) * kTilePad + lane;
        for (int r0 = rbeg; r0 < rend; r0 += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (chan && r0 + u < rend) ? rbt[(size_t)(r0 + u) * dm.C] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u;
                if (r < rend) {
                    while (r >= next) {                          // close voxel vi (possibly empty ones too)
                        cells[vi * kTilePad] = cells[vi * kTilePad] * discount + acc;   // stp3.py:296
                        acc = 0.f;
                        ++vi;
                        next = __builtin_amdgcn_readlane(bound, vi + 1);
                    }
                    acc += x[u];
                }
            }
        }
        for (; vi < 16; ++vi) {                                  // the open voxel and the empty tail
            cells[vi * kTilePad] = cells[vi * kTilePad] * discount + acc;
            acc = 0.f;
        }
        __syncthreads();
        // transposed store: lane = voxel, 16 channel planes per wave, 256-B rows
        const int v = v0 + lane;
        if (v < dm.V) {
            for (int ci = 0; ci < 16; ++ci) {
                const int c = wv * 16 + ci;
                if (c < dm.C) bev[((size_t)bt * dm.C + c) * dm.V + v] = tile[lane * kTilePad + c];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K6: backward
// ------------------------------------------------------------------------------------------
// (a) G_t = sum_{t'>=t} discount^(t'-t) dL/dout[b][t'], transposed to voxel-major [V][C] so that
//     the gather in (b) reads one 256-B row per point.
__global__ __launch_bounds__(256) void bev_grad_accumulate_kernel(Dims dm, const float* __restrict__ grad_bev,
                                                                  float discount, float* __restrict__ gacc) {
    __shared__ float stage[64 * kTilePad];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int v0 = blockIdx.x * kTileV;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int t = dm.T - 1; t >= 0; --t) {
        const int bt = b * dm.T + t;
        const int v = v0 + lane;
        for (int ci = 0; ci < 16; ++ci) {
            const int c = wv * 16 + ci;
            float g = 0.f;
            if (c < dm.C && v < dm.V) g = grad_bev[((size_t)bt * dm.C + c) * dm.V + v];
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

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// sum over the 64 lanes of a wave; the total is returned in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xF>(v);  // row_mirror
    v += dpp_f<0x142, 0xA>(v);  // row_bcast15 -> rows 1,3
    v += dpp_f<0x143, 0xC>(v);  // row_bcast31 -> rows 2,3
    return readlane_f(v, 63);
}

// (b) one wave per image column (bt, n, w) [x a slice of its rows]: lane = channel for feat /
//     dfeat / G and lane = depth bin for prob / vox / dprob.  G[d] holds the voxel-major gradient
//     row of the column's CURRENT run in depth bin d; it is re-fetched (one 256-B row) only when
//     the voxel id changes from one image row to the next, i.e. once per run instead of once per
//     frustum point.  dprob[d] = <feat, G[d]>, dfeat += prob[d] * G[d], then the softmax backward
//     dlogit = prob * (dprob - sum_d prob*dprob) fused per pixel.
template <int DCAP>
__global__ __launch_bounds__(256) void lift_splat_bwd_kernel(Dims dm, int hsplit, const float* __restrict__ gacc,
                                                             const float* __restrict__ feat,
                                                             const float* __restrict__ prob,
                                                             const int32_t* __restrict__ vox_pm,
                                                             float* __restrict__ grad_feat,
                                                             float* __restrict__ grad_logits) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t task = (int64_t)blockIdx.x * 4 + wv;
    if (task >= (int64_t)dm.BT * dm.NCOL * hsplit) return;
    const int hs = (int)(task % hsplit);
    const int64_t tc = task / hsplit;
    const int col = (int)(tc % dm.NCOL), bt = (int)(tc / dm.NCOL);
    const int n = col / dm.fW, w = col - n * dm.fW;
    const int hlen = (dm.fH + hsplit - 1) / hsplit;
    const int h_lo = hs * hlen, h_hi = min(dm.fH, h_lo + hlen);
    const float* g_bt = gacc + (size_t)bt * dm.V * dm.C + lane;
    const bool chan = lane < dm.C;
    const bool bin = lane < dm.D;
    float G[DCAP];
#pragma unroll
    for (int d = 0; d < DCAP; ++d) G[d] = 0.f;
    int curv = -1;
    for (int h = h_lo; h < h_hi; ++h) {
        const size_t gp = (size_t)bt * dm.NPIX + ((size_t)n * dm.fH + h) * dm.fW + w;
        const float f = chan ? feat[gp * dm.C + lane] : 0.f;
        const float pr = bin ? prob[gp * dm.D + lane] : 0.f;
        const int vx = bin ? vox_pm[gp * dm.D + lane] : -1;
        const unsigned long long chg = __ballot(vx != curv);
        curv = vx;
        if (chg) {
#pragma unroll
            for (int d = 0; d < DCAP; ++d) {
                if ((chg >> d) & 1ull) {
                    const int v = __builtin_amdgcn_readlane(vx, d);
                    G[d] = (v >= 0 && chan) ? g_bt[(size_t)v * dm.C] : 0.f;
                }
            }
        }
        float dfeat = 0.f, dprob = 0.f;
#pragma unroll
        for (int d = 0; d < DCAP; ++d) {
            if (d < dm.D) {
                dfeat = fmaf(readlane_f(pr, d), G[d], dfeat);
                const float sd = wave_sum(f * G[d]);
                dprob = (lane == d) ? sd : dprob;
            }
        }
        const float sdot = wave_sum(pr * dprob);
        if (chan) grad_feat[gp * dm.C + lane] = dfeat;
        if (bin) grad_logits[gp * dm.D + lane] = pr * (dprob - sdot);
    }
}

// (b') EXPERIMENTAL (STP3_LIFT_BWD=mfma): the same backward as two small fp32 GEMMs per image column
//      on the matrix cores.  With the column's runs r = (depth bin d_r, rows [h0_r, h1_r), voxel v_r):
//          Ghat[r][c] = G[v_r][c]                        (one 256-B row fetched per run)
//          Phat[h][r] = prob[h][d_r] if h0_r <= h < h1_r else 0
//          dfeat[h][c]   = sum_r Phat[h][r] * Ghat[r][c]               ([fH x R] x [R x C])
//          dprob[h][d_r] = sum_c feat[h][c] * Ghat[r][c], h in the run ([fH x C] x [C x R], masked)
//      One block (4 waves) per column; wave w owns the 16-channel slice w of dfeat and every
//      fourth 16-run tile of dprob.  Runs are enumerated in the block from the voxel ids (bin-major,
//      the order the forward plan uses), processed in chunks of kBwdRunCap.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBwdRunCap = 128;  // runs per chunk (rows of Ghat resident in LDS)
constexpr int kBwdRows = 32;     // image rows covered by the two 16-row MFMA tiles
constexpr int kBwdStride = 66;   // row stride (floats) of the 64-wide LDS tiles: 2h + k spreads the banks

inline size_t lift_bwd_mfma_lds(int Dp) {
    return ((size_t)(kBwdRows + kBwdRunCap) * kBwdStride + 3 * (size_t)kBwdRows * Dp + 2 * kBwdRunCap + 64) *
           sizeof(float);
}

__global__ __launch_bounds__(256) void lift_bwd_mfma_kernel(Dims dm, int Dp, const float* __restrict__ gacc,
                                                            const float* __restrict__ feat,
                                                            const float* __restrict__ prob,
                                                            const int32_t* __restrict__ vox_pm,
                                                            float* __restrict__ grad_feat,
                                                            float* __restrict__ grad_logits) {
    extern __shared__ float smem[];
    float* fcol = smem;                                      // [32][66]  features of the column
    float* ghat = fcol + kBwdRows * kBwdStride;              // [128][66] gradient row of each run
    float* pcol = ghat + kBwdRunCap * kBwdStride;            // [32][Dp]  depth probabilities
    float* tcol = pcol + kBwdRows * Dp;                      // [32][Dp]  dL/dprob
    int* vcol = (int*)(tcol + kBwdRows * Dp);                // [32][Dp]  voxel ids
    int* rdesc = vcol + kBwdRows * Dp;                       // [128] d | h0 << 8 | h1 << 16
    int* rvox = rdesc + kBwdRunCap;                          // [128] voxel id of the run (-1: padding)
    int* rcnt = rvox + kBwdRunCap;                           // [64]  runs per depth bin

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = blockIdx.x, bt = blockIdx.y;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const size_t pix0 = (size_t)bt * dm.NPIX + (size_t)n * dm.fH * dm.fW + w;  // + h * fW

    // ---- stage the column (rows >= fH and channels >= C are zero) ----
#pragma unroll
    for (int i = 0; i < kBwdRows / 4; ++i) {
        const int h = wv + 4 * i;
        const size_t gp = pix0 + (size_t)h * dm.fW;
        const bool row = h < dm.fH;
        fcol[h * kBwdStride + lane] = (row && lane < dm.C) ? feat[gp * dm.C + lane] : 0.f;
        if (lane < Dp) {
            const bool ok = row && lane < dm.D;
            pcol[h * Dp + lane] = ok ? prob[gp * dm.D + lane] : 0.f;
            vcol[h * Dp + lane] = ok ? vox_pm[gp * dm.D + lane] : -1;
            tcol[h * Dp + lane] = 0.f;
        }
    }
    __syncthreads();

    // ---- count the runs of every depth bin (thread = bin) ----
    if (tid < 64) {
        int cnt = 0;
        if (tid < dm.D) {
            int prev = -1;
            for (int h = 0; h < dm.fH; ++h) {
                const int v = vcol[h * Dp + tid];
                cnt += (v >= 0 && v != prev) ? 1 : 0;
                prev = v;
            }
        }
        rcnt[tid] = cnt;
    }
    __syncthreads();
    int my_off = 0, total = 0;
    for (int d = 0; d < dm.D; ++d) {
        const int c = rcnt[d];
        my_off += (d < tid) ? c : 0;
        total += c;
    }

    const int li = lane & 15, kk = lane >> 4;
    f32x4 dacc[2];
    dacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    dacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = 0; base < total; base += kBwdRunCap) {
        const int rc = min(kBwdRunCap, total - base);
        const int rc16 = (rc + 15) & ~15;
        // ---- describe the runs of this chunk ----
        if (tid < dm.D) {
            int idx = my_off - base, prev = -1, h0 = 0;
            for (int h = 0; h <= dm.fH; ++h) {
                const int v = h < dm.fH ? vcol[h * Dp + tid] : -1;
                if (v != prev) {
                    if (prev >= 0) {
                        if (idx >= 0 && idx < kBwdRunCap) {
                            rdesc[idx] = tid | (h0 << 8) | (h << 16);
                            rvox[idx] = prev;
                        }
                        ++idx;
                    }
                    h0 = h;
                    prev = v;
                }
            }
        }
        for (int idx = rc + tid; idx < rc16; idx += 256) {
            rdesc[idx] = 0;  // empty row range
            rvox[idx] = -1;
        }
        __syncthreads();
        // ---- Ghat: one gradient row per run, 8 rows in flight per wave ----
        for (int r0 = wv * 8; r0 < rc16; r0 += 32) {
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0 + j;
                const int v = r < rc16 ? rvox[r] : -1;
                g[j] = (v >= 0 && lane < dm.C) ? gacc[((size_t)bt * dm.V + v) * dm.C + lane] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (r0 + j < rc16) ghat[(r0 + j) * kBwdStride + lane] = g[j];
        }
        __syncthreads();
        // ---- dfeat += Phat x Ghat (wave = 16-channel slice) ----
        if (wv * 16 < dm.C) {
            for (int k0 = 0; k0 < rc16; k0 += 4) {
                const int r = k0 + kk;
                const int desc = rdesc[r];
                const int d = desc & 255, h0 = (desc >> 8) & 255, h1 = desc >> 16;
                const float b = ghat[r * kBwdStride + wv * 16 + li];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int h = m * 16 + li;
                    const float a = (h >= h0 && h < h1) ? pcol[h * Dp + d] : 0.f;
                    dacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, dacc[m], 0, 0, 0);
                }
            }
        }
        // ---- dprob = feat x Ghat^T on the runs' row ranges (wave = every fourth 16-run tile) ----
        for (int rt = wv; rt * 16 < rc16; rt += 4) {
            f32x4 pacc[2];
            pacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            pacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const float b = ghat[(rt * 16 + li) * kBwdStride + k0 + kk];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float a = fcol[(m * 16 + li) * kBwdStride + k0 + kk];
                    pacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, pacc[m], 0, 0, 0);
                }
            }
            const int desc = rdesc[rt * 16 + li];
            const int d = desc & 255, h0 = (desc >> 8) & 255, h1 = desc >> 16;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int h = m * 16 + kk * 4 + q;
                    if (h >= h0 && h < h1) tcol[h * Dp + d] = pacc[m][q];
                }
        }
        __syncthreads();
    }

    // ---- dfeat out: lane holds rows 4*kk + q of tile m, channel 16*wv + li ----
    if (wv * 16 < dm.C) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int h = m * 16 + kk * 4 + q;
                if (h < dm.fH) grad_feat[(pix0 + (size_t)h * dm.fW) * dm.C + wv * 16 + li] = dacc[m][q];
            }
    }
    // ---- softmax backward per pixel: dlogit = p * (dprob - sum_d p * dprob) ----
    for (int h = wv; h < dm.fH; h += 4) {
        const bool bin = lane < dm.D;
        const float pr = bin ? pcol[h * Dp + lane] : 0.f;
        const float dp = bin ? tcol[h * Dp + lane] : 0.f;
        const float sdot = wave_sum(pr * dp);
        if (bin) grad_logits[(pix0 + (size_t)h * dm.fW) * dm.D + lane] = pr * (dp - sdot);
    }
}

// (c) EXPERIMENTAL (STP3_LIFT_FWD=mfma): stage 1 of the forward on the fp32 matrix cores.  With the column's runs
//     r = (depth bin d_r, rows [h0_r, h1_r)) the run vectors are one small GEMM per image column,
//         R[r][c] = sum_h Phat[r][h] * feat[h][c],   Phat[r][h] = prob[h][d_r] if h0_r <= h < h1_r else 0
//     ([runs x fH] x [fH x C]; v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain over h in ascending order, the
//     same order the lane-per-bin kernel adds in).  Wave w owns the 16-channel slice w and walks the 16-run tiles;
//     B (the feature rows) is loaded into registers once per column, A is built from the run descriptors.
//     The column's run ids are contiguous in the plan, their destination rows are staged with one coalesced read.
constexpr int kFwdRows = 32;      // image rows covered by the 8 k-steps
constexpr int kFwdStrideF = 80;   // feature row stride in LDS: 16-lane groups of consecutive k land on disjoint banks

inline size_t lift_runs_mfma_lds(int fH, int Dp) {
    const size_t cap = (((size_t)fH * Dp) + 15) & ~(size_t)15;               // worst case: every point its own run
    return ((size_t)kFwdRows * kFwdStrideF + 2 * (size_t)kFwdRows * Dp + 2 * cap) * sizeof(float);
}

__global__ __launch_bounds__(256) void lift_runs_mfma_kernel(Dims dm, int Dp, const float* __restrict__ feat,
                                                             const float* __restrict__ prob,
                                                             const int32_t* __restrict__ vox_pm,
                                                             const int32_t* __restrict__ run_base,
                                                             const int32_t* __restrict__ dest,
                                                             float* __restrict__ runs) {
    extern __shared__ float smem[];
    float* fcol = smem;                                              // [32][80]
    float* pcol = fcol + kFwdRows * kFwdStrideF;                     // [32][Dp]
    int* vcol = (int*)(pcol + kFwdRows * Dp);                        // [32][Dp]
    const int cap = (dm.fH * Dp + 15) & ~15;
    int* rdesc = vcol + kFwdRows * Dp;                               // [cap] d | h0 << 8 | h1 << 16
    int* rdst = rdesc + cap;                                         // [cap] row of the run in `runs` (-1: padding)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = blockIdx.x, bt = blockIdx.y;
    const int n = col / dm.fW, w = col - n * dm.fW;
    const size_t pix0 = (size_t)bt * dm.NPIX + (size_t)n * dm.fH * dm.fW + w;

    // ---- stage the column (rows >= fH, channels >= C, bins >= D are zero / -1) ----
#pragma unroll
    for (int i = 0; i < kFwdRows / 4; ++i) {
        const int h = wv + 4 * i;
        const size_t gp = pix0 + (size_t)h * dm.fW;
        const bool row = h < dm.fH;
        fcol[h * kFwdStrideF + lane] = (row && lane < dm.C) ? feat[gp * dm.C + lane] : 0.f;
        if (lane < Dp) {
            const bool ok = row && lane < dm.D;
            pcol[h * Dp + lane] = ok ? prob[gp * dm.D + lane] : 0.f;
            vcol[h * Dp + lane] = ok ? vox_pm[gp * dm.D + lane] : -1;
        }
    }
    const int32_t* rb_col = run_base + (size_t)bt * (dm.NQ + 1) + (size_t)col * dm.D;
    const int col_base = rb_col[0], col_runs = rb_col[dm.D] - col_base;
    const int runs16 = (col_runs + 15) & ~15;
    const int32_t* dst = dest + (size_t)bt * dm.P + col_base;
    for (int i = tid; i < runs16; i += 256) {
        rdst[i] = i < col_runs ? dst[i] : -1;
        if (i >= col_runs) rdesc[i] = 0;                             // empty row range
    }
    __syncthreads();
    // ---- describe the runs: thread = depth bin, ids in plan order (bin-major, rows ascending) ----
    if (tid < dm.D) {
        int idx = rb_col[tid] - col_base, prev = -1, h0 = 0;
        for (int h = 0; h <= dm.fH; ++h) {
            const int v = h < dm.fH ? vcol[h * Dp + tid] : -1;
            if (v != prev) {
                if (prev >= 0) rdesc[idx++] = tid | (h0 << 8) | (h << 16);
                h0 = h;
                prev = v;
            }
        }
    }
    __syncthreads();
    if (wv * 16 >= dm.C) return;

    // ---- R = Phat x F: B fragments (feature rows of this wave's 16 channels) live in registers ----
    const int li = lane & 15, kk = lane >> 4;
    float bq[kFwdRows / 4];
#pragma unroll
    for (int ks = 0; ks < kFwdRows / 4; ++ks) bq[ks] = fcol[(4 * ks + kk) * kFwdStrideF + wv * 16 + li];
    float* out = runs + (size_t)bt * dm.P * dm.C + wv * 16 + li;
    for (int r0 = 0; r0 < runs16; r0 += 16) {
        const int desc = rdesc[r0 + li];
        const int d = desc & 255, h0 = (desc >> 8) & 255, h1 = desc >> 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kFwdRows / 4; ++ks) {
            const int h = 4 * ks + kk;
            const float a = (h >= h0 && h < h1) ? pcol[h * Dp + d] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[ks], acc, 0, 0, 0);
        }
        // lane holds runs r0 + 4*kk + q (q = 0..3), channel 16*wv + li
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rdst[r0 + 4 * kk + q];
            if (row >= 0) out[(size_t)row * dm.C] = acc[q];
        }
    }
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char* stp3_version(void) { return "stp3hip 0.1 gfx950"; }

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

int stp3_lift_plan_bytes(const stp3_lift_dims* dims, size_t* bytes) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = plan_bytes(dm);
    return STP3_OK;
}

int stp3_lift_workspace_bytes(const stp3_lift_dims* dims, size_t* bytes) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = (size_t)dm.BT * dm.P * dm.C * sizeof(float);   // worst case: every frustum point its own run
    return STP3_OK;
}

int stp3_lift_plan_build(const stp3_lift_dims* dims, const int32_t* vox_pm, int32_t* counts, void* plan,
                         size_t plan_size, int deterministic, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!vox_pm || !counts || !plan) return STP3_EINVAL;
    if (plan_size < plan_bytes(dm)) return STP3_ENOSPACE;
    PlanView pv = plan_view(dm, plan);
    hipStream_t s = (hipStream_t)stream;
    const dim3 qgrid((dm.NQ + 255) / 256, dm.BT);
    hipLaunchKernelGGL(run_count_kernel, qgrid, dim3(256), 0, s, dm, vox_pm, pv.run_base, counts);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(dm.BT), dim3(1024), 0, s, dm.NQ, dm.NQ + 1, pv.run_base, pv.run_base, 0);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(dm.BT), dim3(1024), 0, s, dm.V, dm.V, counts, pv.vox_off, 1);
    hipLaunchKernelGGL(run_fill_kernel, qgrid, dim3(256), 0, s, dm, vox_pm, pv.run_base, pv.vox_off, counts, pv.list);
    if (deterministic) {
        int64_t items = (int64_t)dm.BT * dm.V;
        int blocks = (int)((items + 1) / 2 < 4096 ? (items + 1) / 2 : 4096);
        hipLaunchKernelGGL(plan_sort_kernel, dim3(blocks), dim3(128), 0, s, dm, pv.vox_off, pv.list);
    }
    hipLaunchKernelGGL(plan_invert_kernel, dim3(256, dm.BT), dim3(256), 0, s, dm, pv.vox_off, pv.list, pv.dest);
    return launch_status();
}

int stp3_depth_softmax(const stp3_lift_dims* dims, const float* logits, float* prob, void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!logits || !prob) return STP3_EINVAL;
    if (dm.D > 128) return STP3_EUNSUP;
    const int64_t npix = (int64_t)dm.BT * dm.NPIX;
    hipStream_t s = (hipStream_t)stream;
    if (dm.D <= 32) {
        hipLaunchKernelGGL(depth_softmax_kernel<8>, dim3((unsigned)((npix + 31) / 32)), dim3(256), 0, s, npix, dm.D,
                           logits, prob);
    } else if (dm.D <= 64) {
        hipLaunchKernelGGL(depth_softmax_kernel<16>, dim3((unsigned)((npix + 15) / 16)), dim3(256), 0, s, npix, dm.D,
                           logits, prob);
    } else {
        hipLaunchKernelGGL(depth_softmax_kernel<32>, dim3((unsigned)((npix + 7) / 8)), dim3(256), 0, s, npix, dm.D,
                           logits, prob);
    }
    return launch_status();
}

int stp3_lift_splat_fwd(const stp3_lift_dims* dims, const float* feat, const float* prob, const int32_t* vox_pm,
                        const void* plan, float discount, void* workspace, size_t workspace_bytes, float* bev,
                        void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!feat || !prob || !vox_pm || !plan || !workspace || !bev) return STP3_EINVAL;
    if (dm.Z != 1 || dm.C > 64 || dm.D > 128) return STP3_EUNSUP;  // stp3.py:297-299 squeezes Z; lane = channel
    if (workspace_bytes < (size_t)dm.BT * dm.P * dm.C * sizeof(float)) return STP3_ENOSPACE;
    PlanView pv = plan_view(dm, const_cast<void*>(plan));
    hipStream_t s = (hipStream_t)stream;
    const int ndg = (dm.D + 7) / 8;                        // waves per column: 8 depth bins each
    const int Dp = ndg * 8;
    const size_t lds = ((size_t)dm.fH * 64 + 3 * (size_t)dm.fH * Dp) * sizeof(float);
    // experimental matrix-core variant of stage 1, opt-in (see lift_runs_mfma_kernel)
    static const bool want_mfma = [] {
        const char* e = getenv("STP3_LIFT_FWD");
        return e && !strcmp(e, "mfma");
    }();
    if (want_mfma && dm.fH <= kFwdRows && dm.C % 16 == 0 && dm.D <= 64 && dm.BT <= 65535) {
        const int Dq = dm.D | 1;
        const size_t lds_m = lift_runs_mfma_lds(dm.fH, Dq);
        if (lds_m <= 64 * 1024) {
            hipLaunchKernelGGL(lift_runs_mfma_kernel, dim3(dm.NCOL, dm.BT), dim3(256), lds_m, s, dm, Dq, feat, prob, vox_pm,
                               pv.run_base, pv.dest, (float*)workspace);
            hipLaunchKernelGGL(lift_gather_kernel, dim3((dm.V + kTileV - 1) / kTileV, dm.B), dim3(256), 0, s, dm,
                               (const float*)workspace, pv.vox_off, discount, bev);
            return launch_status();
        }
    }
    if (lds > 160 * 1024) return STP3_EUNSUP;
    if (lds > 64 * 1024) {
        // tall feature maps (BASELINE configs[4]: 112 rows x 64 bins = 112 KB per column): above the default
        // dynamic-LDS limit of a launch, raise it for this kernel (gfx950 has 160 KB per workgroup)
        const void* fn = dm.C % 8 == 0 ? reinterpret_cast<const void*>(&lift_runs_kernel<true>)
                                       : reinterpret_cast<const void*>(&lift_runs_kernel<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
    }
    if (dm.C % 8 == 0)
        hipLaunchKernelGGL((lift_runs_kernel<true>), dim3(dm.NCOL, dm.BT), dim3(64 * ndg), lds, s, dm, Dp, feat, prob,
                           vox_pm, pv.run_base, pv.dest, (float*)workspace);
    else
        hipLaunchKernelGGL((lift_runs_kernel<false>), dim3(dm.NCOL, dm.BT), dim3(64 * ndg), lds, s, dm, Dp, feat, prob,
                           vox_pm, pv.run_base, pv.dest, (float*)workspace);
    hipLaunchKernelGGL(lift_gather_kernel, dim3((dm.V + kTileV - 1) / kTileV, dm.B), dim3(256), 0, s, dm,
                       (const float*)workspace, pv.vox_off, discount, bev);
    return launch_status();
}

int stp3_lift_splat_bwd(const stp3_lift_dims* dims, const float* grad_bev, const float* feat, const float* prob,
                        const int32_t* vox_pm, float discount, float* gacc, float* grad_feat, float* grad_logits,
                        void* stream) {
    Dims dm;
    int rc = check_dims(dims, &dm);
    if (rc) return rc;
    if (!grad_bev || !feat || !prob || !vox_pm || !gacc || !grad_feat || !grad_logits) return STP3_EINVAL;
    if (dm.Z != 1 || dm.C > 64 || dm.D > 64) return STP3_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bev_grad_accumulate_kernel, dim3((dm.V + kTileV - 1) / kTileV, dm.B), dim3(256), 0, s, dm,
                       grad_bev, discount, gacc);
    // experimental matrix-core variant, opt-in (see lift_bwd_mfma_kernel)
    static const bool want_mfma = [] {
        const char* e = getenv("STP3_LIFT_BWD");
        return e && !strcmp(e, "mfma");
    }();
    if (want_mfma && dm.fH <= kBwdRows && dm.C % 16 == 0 && dm.BT <= 65535) {
        const int Dp = dm.D | 1;
        const size_t lds = lift_bwd_mfma_lds(Dp);
        if (lds <= 64 * 1024) {
            hipLaunchKernelGGL(lift_bwd_mfma_kernel, dim3(dm.NCOL, dm.BT), dim3(256), lds, s, dm, Dp, gacc, feat, prob,
                               vox_pm, grad_feat, grad_logits);
            return launch_status();
        }
    }
    // enough waves to fill the chip: split the rows of a column when there are few columns
    const int64_t cols = (int64_t)dm.BT * dm.NCOL;
    int hsplit = (int)((8192 + cols - 1) / cols);
    const int max_split = dm.fH / 8 > 0 ? dm.fH / 8 : 1;
    if (hsplit > max_split) hsplit = max_split;
    if (hsplit < 1) hsplit = 1;
    const int64_t tasks = cols * hsplit;
    const dim3 grid((unsigned)((tasks + 3) / 4));
    if (dm.D <= 48)
        hipLaunchKernelGGL(lift_splat_bwd_kernel<48>, grid, dim3(256), 0, s, dm, hsplit, gacc, feat, prob, vox_pm,
                           grad_feat, grad_logits);
    else
        hipLaunchKernelGGL(lift_splat_bwd_kernel<64>, grid, dim3(256), 0, s, dm, hsplit, gacc, feat, prob, vox_pm,
                           grad_feat, grad_logits);
    return launch_status();
}

}  // extern "C"
