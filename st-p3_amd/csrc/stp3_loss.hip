// stp3_loss.hip -- the training losses of the perception path and the nearest-neighbour label warp (gfx950).
//
// Replaces, in the timed training step,
//   * stp3/losses.py:43-83 SegmentationLoss, :85-114 HDmapLoss, :116-134 DepthLoss: weighted cross-entropy per pixel
//     (ignore_index), future discount, and the mean of the k largest per-pixel losses of every (sample, frame) row --
//     the reference sorts all 40 000 pixels of a row; only the k-th largest VALUE is needed, found here by a 4-pass
//     radix select on the float bits inside one workgroup;
//   * stp3/losses.py:6-40 SpatialRegressionLoss: L1 / L2 over the channels, future discount, mean over the pixels whose
//     target is not ignore_index;
//   * stp3/utils/geometry.py:196-238 warp_features as used by prepare_future_labels (stp3/trainer.py:254-360):
//     F.affine_grid + F.grid_sample(mode='nearest', padding_mode='zeros', align_corners=False) of the label maps, all
//     frames and all label channels in one launch.
// Written with torch operators a step spends ~150 launches on these (log-softmax, nll, discounts, a multi-block top-k
// with six kernels per call, staged means, per-frame affine grids ...); here 3 + 1 launches per loss and 1 for the warp.
// Everything is float32 arithmetic on float32 / bf16 logits; reductions run in a fixed order (deterministic).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

constexpr int kT = 256;
constexpr int kSelT = 1024;                 // threads of the per-row select workgroup

struct CeDims {
    int rows, P, C, k, ignore_index, bf16;
    long long stride_row, stride_c, stride_p;
};

__device__ __forceinline__ float ld_logit(const void* base, bool bf16, long long off) {
    if (bf16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[off] << 16);
    return reinterpret_cast<const float*>(base)[off];
}

// ---- per-pixel weighted cross-entropy: loss[row][p] = scale[row] * w[y] * (logsumexp(z) - z[y]), 0 where y == ignore ----
__global__ __launch_bounds__(kT) void ce_pixel_kernel(CeDims d, const void* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      const float* __restrict__ weight, const float* __restrict__ row_scale,
                                                      float* __restrict__ loss) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= (long long)d.rows * d.P) return;
    const int row = (int)(i / d.P), p = (int)(i - (long long)row * d.P);
    const int64_t y = labels[i];
    float out = 0.f;
    if (y != d.ignore_index && y >= 0 && y < d.C) {
        const long long base = (long long)row * d.stride_row + (long long)p * d.stride_p;
        float m = -INFINITY;
        for (int c = 0; c < d.C; ++c) m = fmaxf(m, ld_logit(logits, d.bf16, base + c * d.stride_c));
        float s = 0.f;
        for (int c = 0; c < d.C; ++c) s += expf(ld_logit(logits, d.bf16, base + c * d.stride_c) - m);
        const float zy = ld_logit(logits, d.bf16, base + y * d.stride_c);
        out = (weight ? weight[y] : 1.f) * ((m - zy) + logf(s)) * (row_scale ? row_scale[row] : 1.f);
    }
    loss[i] = out;
}

// block-wide sum in a fixed order: a tree over LDS (red: blockDim.x doubles)
__device__ __forceinline__ double block_sum(double v, double* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

// ---- per row: the k-th largest loss tau (radix select on the bits: losses are >= 0, so unsigned order = float order),
// rowsum = sum of the k largest = sum_{l > tau} l + (k - #{l > tau}) * tau, and the share of every tie at tau in the
// gradient, (k - #{l > tau}) / #{l == tau}.  k >= P: no selection, rowsum = sum of the row.
// sel[row] = {tau, tie_share};  grid = rows, kSelT threads.
// Six rows of 40 000 losses are six workgroups: what the kernel costs is its own serial chain.  Round 6 took three links out of
// it (52 -> ~15 us per call, five calls per step): the row is read ONCE into registers (kSelOwn values per thread: rows up to
// 40 960 pixels -- the 200 x 200 BEV; longer rows re-read it per pass as before), the bin that holds the need-th largest is
// found by one wave (four bins per lane, a suffix scan over the lanes) instead of one thread walking 256 LDS words four times,
// and the two closing block sums share their tree.
constexpr int kSelOwn = 40;
template <bool CACHED>
__global__ __launch_bounds__(kSelT) void topk_select_kernel(int P, int k, const float* __restrict__ loss, float* __restrict__ sel,
                                                            double* __restrict__ rowsum) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need;
    __shared__ double red[kSelT];
    __shared__ double red2[kSelT];
    const int row = blockIdx.x;
    const float* l = loss + (size_t)row * P;
    if (k >= P || k <= 0) {
        double s = 0.0;
        for (int i = threadIdx.x; i < P; i += kSelT) s += (double)l[i];
        s = block_sum(s, red);
        if (threadIdx.x == 0) {
            rowsum[row] = s;
            sel[2 * row] = -1.f;                       // every loss is > tau
            sel[2 * row + 1] = 0.f;
        }
        return;
    }
    // this thread's elements: i = threadIdx.x + j * kSelT
    const int own = CACHED ? kSelOwn : (P + kSelT - 1) / kSelT;
    float v[kSelOwn];
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < kSelOwn; ++j) {
            const int i = threadIdx.x + j * kSelT;
            v[j] = i < P ? l[i] : -1.f;                 // (beyond the row: its sign bit keeps it out of every pass; neither > nor == tau)
        }
    }
    unsigned prefix = 0, need = (unsigned)k;            // bits fixed so far; how many of the largest are still to be found
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        // (pass 0 only asks for a clear sign bit: losses are >= +0; the -1 that pads a thread's registers beyond the row is not one)
        const unsigned mask_hi = pass == 0 ? 0x80000000u : (0xffffffffu << (shift + 8));
        // run-length aggregation in registers: the losses of a row share their leading bytes (one or two exponents), so a
        // plain atomic per element serialises 40 000 increments on one or two LDS words (64 us per call measured); a thread
        // counts its consecutive equal digits and issues one atomic per run
        unsigned cur = 0xffffffffu, cnt = 0;
        auto take = [&](unsigned u) {
            if ((u & mask_hi) != prefix) return;
            const unsigned b = (u >> shift) & 255u;
            if (b == cur) { ++cnt; return; }
            if (cnt) atomicAdd(&hist[cur], cnt);                                         // integer counts: order-free
            cur = b;
            cnt = 1;
        };
        if (CACHED) {
#pragma unroll
            for (int j = 0; j < kSelOwn; ++j) take(__float_as_uint(v[j]));
        } else {
            for (int j = 0; j < own; ++j) {
                const int i = threadIdx.x + j * kSelT;
                if (i < P) take(__float_as_uint(l[i]));
            }
        }
        if (cnt) atomicAdd(&hist[cur], cnt);
        __syncthreads();
        // the bin that holds the need-th largest, from the top: the highest b with sum_{b' >= b} hist[b'] >= need (bin 0 when
        // even that sum falls short), and how many of ITS elements are still needed
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const unsigned tot = h0 + h1 + h2 + h3;
            unsigned suf = tot;                           // counts of this lane's bins and every higher lane's
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl(suf, min(lane + off, 63), 64);
                if (lane + off < 64) suf += o;
            }
            const unsigned above = suf - tot;
            const bool mine = (above < need && need <= suf) || (lane == 0 && suf < need);
            if (mine) {
                unsigned n = need - above, b = 4 * lane + 3;
                if (h3 < n) {
                    n -= h3; --b;
                    if (h2 < n) {
                        n -= h2; --b;
                        if (h1 < n) { n -= h1; --b; }
                    }
                }
                s_prefix = prefix | (b << shift);
                s_need = n;
            }
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
    }
    const float tau = __uint_as_float(prefix);           // `need` of the elements equal to tau belong to the top k
    double s = 0.0;
    unsigned ties = 0;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < kSelOwn; ++j) {
            if (v[j] > tau) s += (double)v[j];
            ties += v[j] == tau ? 1u : 0u;
        }
    } else {
        for (int i = threadIdx.x; i < P; i += kSelT) {
            const float x = l[i];
            if (x > tau) s += (double)x;
            ties += x == tau ? 1u : 0u;
        }
    }
    // both block sums through one tree (fixed order)
    red[threadIdx.x] = s;
    red2[threadIdx.x] = (double)ties;
    __syncthreads();
    for (int st = kSelT >> 1; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[threadIdx.x] += red[threadIdx.x + st];
            red2[threadIdx.x] += red2[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tcount = red2[0];
        rowsum[row] = red[0] + (double)need * (double)tau;
        sel[2 * row] = tau;
        sel[2 * row + 1] = tcount > 0.0 ? (float)((double)need / tcount) : 0.f;
    }
}

// out[0] = scale * sum_r rowsum[r] (+ out[0] when accumulate): one thread, rows in ascending order
__global__ void ce_finalize_kernel(int rows, const double* __restrict__ rowsum, double scale, int accumulate, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int r = 0; r < rows; ++r) s += rowsum[r];
    const float v = (float)(s * scale);
    out[0] = accumulate ? out[0] + v : v;
}

// ---- backward: dlogits[row][p][c] = gout * coef * take(p) * scale[row] * w[y] * (softmax_c - [c == y]) ---------------
__global__ __launch_bounds__(kT) void ce_backward_kernel(CeDims d, const void* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ weight, const float* __restrict__ row_scale,
                                                         const float* __restrict__ loss, const float* __restrict__ sel,
                                                         const float* __restrict__ gout, float coef, void* __restrict__ dlogits) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= (long long)d.rows * d.P) return;
    const int row = (int)(i / d.P), p = (int)(i - (long long)row * d.P);
    const int64_t y = labels[i];
    const long long base = (long long)row * d.stride_row + (long long)p * d.stride_p;
    float g = 0.f;
    if (y != d.ignore_index && y >= 0 && y < d.C) {
        const float v = loss[i], tau = sel[2 * row];
        const float take = v > tau ? 1.f : (v == tau ? sel[2 * row + 1] : 0.f);
        g = gout[0] * coef * take * (row_scale ? row_scale[row] : 1.f) * (weight ? weight[y] : 1.f);
    }
    float m = -INFINITY, s = 0.f;
    if (g != 0.f) {
        for (int c = 0; c < d.C; ++c) m = fmaxf(m, ld_logit(logits, d.bf16, base + c * d.stride_c));
        for (int c = 0; c < d.C; ++c) s += expf(ld_logit(logits, d.bf16, base + c * d.stride_c) - m);
    }
    for (int c = 0; c < d.C; ++c) {
        float v = 0.f;
        if (g != 0.f) v = g * (expf(ld_logit(logits, d.bf16, base + c * d.stride_c) - m) / s - (c == y ? 1.f : 0.f));
        if (d.bf16) reinterpret_cast<uint16_t*>(dlogits)[base + c * d.stride_c] = (uint16_t)pack_bf16(v, 0.f);
        else reinterpret_cast<float*>(dlogits)[base + c * d.stride_c] = v;
    }
}

// ---- masked regression loss -------------------------------------------------------------------------------------
// pred, target [rows][C][P] (contiguous); per pixel l = scale[row] * sum_c |d| (norm 1) or d^2 (norm 2) where
// target[row][0][p] != ignore; partial[block] = (sum l, count) ; grid-stride, fixed order.
__global__ __launch_bounds__(kT) void reg_loss_kernel(int rows, int C, int P, int norm, float ignore, int bf16,
                                                      const void* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ row_scale, double* __restrict__ partial) {
    __shared__ double red[kT];
    double s = 0.0, cnt = 0.0;
    const long long total = (long long)rows * P;
    for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
        const int row = (int)(i / P), p = (int)(i - (long long)row * P);
        const long long base = (long long)row * C * P + p;
        if (target[base] != ignore) {
            float l = 0.f;
            for (int c = 0; c < C; ++c) {
                const float dlt = ld_logit(pred, bf16 != 0, base + (long long)c * P) - target[base + (long long)c * P];
                l += norm == 1 ? fabsf(dlt) : dlt * dlt;
            }
            s += (double)(l * (row_scale ? row_scale[row] : 1.f));
            cnt += 1.0;
        }
    }
    s = block_sum(s, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = cnt;
    }
}

// out[0] = sum / max(count, 1); out[1] = count  (one workgroup: strided partial sums, then a fixed tree)
__global__ __launch_bounds__(kT) void reg_finalize_kernel(int parts, const double* __restrict__ partial, float* __restrict__ out) {
    __shared__ double red[kT];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < parts; i += kT) { s += partial[2 * i]; c += partial[2 * i + 1]; }
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        out[0] = (float)(s / (c > 0.0 ? c : 1.0));
        out[1] = (float)c;
    }
}

__global__ __launch_bounds__(kT) void reg_backward_kernel(int rows, int C, int P, int norm, float ignore, int bf16,
                                                          const void* __restrict__ pred, const float* __restrict__ target,
                                                          const float* __restrict__ row_scale, const float* __restrict__ stat,
                                                          const float* __restrict__ gout, void* __restrict__ dpred) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= (long long)rows * P) return;
    const int row = (int)(i / P), p = (int)(i - (long long)row * P);
    const long long base = (long long)row * C * P + p;
    const bool on = target[base] != ignore;
    const float g = on ? gout[0] * (row_scale ? row_scale[row] : 1.f) / fmaxf(stat[1], 1.f) : 0.f;
    for (int c = 0; c < C; ++c) {
        const long long o = base + (long long)c * P;
        float v = 0.f;
        if (on) {
            const float dlt = ld_logit(pred, bf16 != 0, o) - target[o];
            v = norm == 1 ? (dlt > 0.f ? g : (dlt < 0.f ? -g : 0.f)) : 2.f * dlt * g;
        }
        if (bf16) reinterpret_cast<uint16_t*>(dpred)[o] = (uint16_t)pack_bf16(v, 0.f);
        else reinterpret_cast<float*>(dpred)[o] = v;
    }
}

// ---- nearest-neighbour affine warp (label maps) ---------------------------------------------------------------------
// x, y [frames][C][H][W] float32; theta [frames][6] row-major 2x3 (F.affine_grid convention), identity[frames] != 0: copy.
__global__ __launch_bounds__(kT) void warp_nearest_kernel(int frames, int C, int H, int W, const float* __restrict__ x,
                                                          const float* __restrict__ theta, const int32_t* __restrict__ identity,
                                                          float* __restrict__ y) {
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;
    const long long plane = (long long)H * W;
    if (i >= (long long)frames * plane) return;
    const int f = (int)(i / plane);
    const int r = (int)(i - (long long)f * plane);
    const int hi = r / W, wi = r - hi * W;
    long long src = -1;
    if (identity && identity[f]) {
        src = r;
    } else {
        const float* th = theta + 6 * f;
        // F.affine_grid(align_corners=False): base coordinates of the pixel centres in [-1, 1]
        const float bx = (2.f * wi + 1.f) / W - 1.f, by = (2.f * hi + 1.f) / H - 1.f;
        const float gx = bx * th[0] + by * th[1] + th[2];
        const float gy = bx * th[3] + by * th[4] + th[5];
        // F.grid_sample(align_corners=False): unnormalise, round half to even, zeros outside
        const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
        const float rx = nearbyintf(ix), ry = nearbyintf(iy);
        if (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) src = (long long)ry * W + (long long)rx;
    }
    for (int c = 0; c < C; ++c) {
        const long long o = ((long long)f * C + c) * plane;
        y[o + r] = src >= 0 ? x[o + src] : 0.f;
    }
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline int to_dims(const stp3_ce_dims* p, CeDims* d) {
    if (!p || p->rows <= 0 || p->P <= 0 || p->C <= 0) return STP3_EINVAL;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if ((int64_t)p->rows * p->P >= (1LL << 31)) return STP3_EUNSUP;
    d->rows = p->rows; d->P = p->P; d->C = p->C; d->k = p->k; d->ignore_index = p->ignore_index;
    d->bf16 = p->dtype == STP3_DTYPE_BF16;
    d->stride_row = p->stride_row; d->stride_c = p->stride_c; d->stride_p = p->stride_p;
    return STP3_OK;
}

constexpr int kRegBlocks = 512;

}  // namespace

extern "C" {

int stp3_ce_topk_workspace_bytes(const stp3_ce_dims* p, size_t* bytes) {
    if (!p || !bytes || p->rows <= 0) return STP3_EINVAL;
    *bytes = (size_t)p->rows * sizeof(double);
    return STP3_OK;
}

int stp3_ce_topk_fwd(const stp3_ce_dims* p, const void* logits, const int64_t* labels, const float* class_weights,
                     const float* row_scale, float* loss_px, float* sel, double out_scale, int32_t accumulate, float* out,
                     void* workspace, size_t workspace_bytes, void* stream) {
    CeDims d;
    int rc = to_dims(p, &d);
    if (rc) return rc;
    if (!logits || !labels || !loss_px || !sel || !out || !workspace) return STP3_EINVAL;
    if (workspace_bytes < (size_t)p->rows * sizeof(double)) return STP3_ENOSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)d.rows * d.P;
    hipLaunchKernelGGL(ce_pixel_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, s, d, logits, labels, class_weights,
                       row_scale, loss_px);
    if ((long long)d.P <= (long long)kSelOwn * kSelT)
        hipLaunchKernelGGL(topk_select_kernel<true>, dim3(d.rows), dim3(kSelT), 0, s, d.P, d.k, (const float*)loss_px, sel,
                           (double*)workspace);
    else
        hipLaunchKernelGGL(topk_select_kernel<false>, dim3(d.rows), dim3(kSelT), 0, s, d.P, d.k, (const float*)loss_px, sel,
                           (double*)workspace);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, s, d.rows, (const double*)workspace, out_scale, (int)accumulate,
                       out);
    return status();
}

int stp3_ce_topk_bwd(const stp3_ce_dims* p, const void* logits, const int64_t* labels, const float* class_weights,
                     const float* row_scale, const float* loss_px, const float* sel, const float* gout, double out_scale,
                     void* dlogits, void* stream) {
    CeDims d;
    int rc = to_dims(p, &d);
    if (rc) return rc;
    if (!logits || !labels || !loss_px || !sel || !gout || !dlogits) return STP3_EINVAL;
    const long long n = (long long)d.rows * d.P;
    hipLaunchKernelGGL(ce_backward_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, (hipStream_t)stream, d, logits, labels,
                       class_weights, row_scale, loss_px, sel, gout, (float)out_scale, dlogits);
    return status();
}

int stp3_reg_loss_workspace_bytes(size_t* bytes) {
    if (!bytes) return STP3_EINVAL;
    *bytes = (size_t)kRegBlocks * 2 * sizeof(double);
    return STP3_OK;
}

int stp3_reg_loss_fwd(int32_t rows, int32_t C, int32_t P, int32_t norm, float ignore_value, int32_t dtype, const void* pred,
                      const float* target, const float* row_scale, float* out, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (rows <= 0 || C <= 0 || P <= 0 || (norm != 1 && norm != 2) || !pred || !target || !out || !workspace) return STP3_EINVAL;
    if (dtype != STP3_DTYPE_F32 && dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (workspace_bytes < (size_t)kRegBlocks * 2 * sizeof(double)) return STP3_ENOSPACE;
    if ((int64_t)rows * C * P >= (1LL << 40)) return STP3_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)rows * P;
    int blocks = (int)((n + kT - 1) / kT);
    if (blocks > kRegBlocks) blocks = kRegBlocks;
    hipLaunchKernelGGL(reg_loss_kernel, dim3(blocks), dim3(kT), 0, s, (int)rows, (int)C, (int)P, (int)norm, ignore_value,
                       dtype == STP3_DTYPE_BF16 ? 1 : 0, pred, target, row_scale, (double*)workspace);
    hipLaunchKernelGGL(reg_finalize_kernel, dim3(1), dim3(kT), 0, s, blocks, (const double*)workspace, out);
    return status();
}

int stp3_reg_loss_bwd(int32_t rows, int32_t C, int32_t P, int32_t norm, float ignore_value, int32_t dtype, const void* pred,
                      const float* target, const float* row_scale, const float* stat, const float* gout, void* dpred,
                      void* stream) {
    if (rows <= 0 || C <= 0 || P <= 0 || (norm != 1 && norm != 2) || !pred || !target || !stat || !gout || !dpred) return STP3_EINVAL;
    if (dtype != STP3_DTYPE_F32 && dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    const long long n = (long long)rows * P;
    hipLaunchKernelGGL(reg_backward_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, (hipStream_t)stream, (int)rows, (int)C,
                       (int)P, (int)norm, ignore_value, dtype == STP3_DTYPE_BF16 ? 1 : 0, pred, target, row_scale, stat, gout, dpred);
    return status();
}

int stp3_warp_nearest(int32_t frames, int32_t C, int32_t H, int32_t W, const float* x, const float* theta,
                      const int32_t* identity, float* y, void* stream) {
    if (frames <= 0 || C <= 0 || H <= 0 || W <= 0 || !x || !theta || !y) return STP3_EINVAL;
    const long long n = (long long)frames * H * W;
    if (n >= (1LL << 31)) return STP3_EUNSUP;
    hipLaunchKernelGGL(warp_nearest_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, (hipStream_t)stream, (int)frames, (int)C,
                       (int)H, (int)W, x, theta, identity, y);
    return status();
}

}  // extern "C"
