// stp3_conv.hip -- bf16 MFMA implicit-GEMM 2-D convolution (NHWC) for gfx950: LDS-tiled, 32x32x16 MFMA.
//
// Replaces the dense nn.Conv2d / frame-folded nn.Conv3d contractions of the reference's hot path:
//   stp3/layers/convolutions.py:183-280 (UpsamplingConcat / UpsamplingAdd / ASPP / DeepLabHead 1x1, 3x3 and
//   dilated 3x3), stp3/layers/temporal.py:252-273, 315-325 (CausalConv3d (2,3,3)/(1,3,3) and 1x1x1, run
//   frame-folded as 2-D convolutions), stp3/models/decoder.py:22-140 (7x7/2 stem, ResNet-18 3x3, heads) and the
//   3x3/2 stem and the 1x1 expand / project convolutions of the EfficientNet MBConv blocks driven by
//   stp3/models/encoder.py:57-97 -- forward and, with the taps flipped, the data gradient.
//
// GEMM view:  Y[m][co] = sum_k X[pixel(m) + tap(k)][ci(k)] * W[co][k],  m = (n, ho, wo),  k = tap * Cin + ci.
// Both operands are K-contiguous in memory (NHWC activations, [Cout][KH][KW][Cin] weights), so a 16-byte piece =
// 8 consecutive k of one row, for either operand.
//
// Workgroup = 4 waves = 128 pixels x BN output channels (BN = 128 or 64), K in steps of 64:
//   * staging: every thread copies 16-byte pieces global -> registers -> LDS (zero for padding taps, rows beyond M /
//     Cout and the K tail); the loads of step s+1 are issued before the MFMAs of step s and written to the other LDS
//     buffer after them: one barrier per K step, global latency behind the matrix work of the whole step;
//   * LDS image: [row][8 pieces of 16 bytes], piece j of row r stored at slot j ^ ((r >> 1) & 7): the ds_read_b128 of
//     an MFMA fragment (32 rows, one piece column) is then conflict-free;
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHT tile as operand A (rows = output channels) and the PIXEL tile as
//     operand B: an accumulator lane holds, for ONE pixel, four consecutive output channels per register quad;
//   * epilogue: the tile goes back through LDS ([pixel][channel], bias added, rounded to the output type) and leaves
//     as 16-byte pieces, 256 contiguous bytes per pixel and BN = 128 bf16 channels; optionally the per-channel sum and
//     sum of squares of the ROUNDED outputs over the tile's pixels (BatchNorm statistics: the separate statistics
//     pass over the convolution output disappears), one partial row per workgroup, reduced deterministically.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct ConvDims {
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int ldx, ldy;
    int out_f32, has_bias;
    int M;          // N * Ho * Wo
    int kchunks;    // (unused)
    int Ktot;       // KH * KW * Cin
};

union Frag {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }

constexpr int kBM = 128;        // pixels per workgroup
constexpr int kBK = 64;         // K per step: 8 pieces of 16 bytes per row

// 16 bytes of zeros in device memory: what a staging thread loads INSTEAD of a padding tap / a row beyond the matrix.
// Selecting the ADDRESS (two v_cndmask) replaces selecting the loaded DATA afterwards (four v_cndmask per piece plus
// the bookkeeping of which pieces were real): the PMC passes of round 3 showed both convolution kernels bound by their
// VALU instruction count (196 VALU instructions beside 8 MFMAs per K-step in the forward kernel, 332 beside 4 in the
// weight gradient), not by the matrix cores or by memory.
__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0u, 0u, 0u, 0u};
typedef stp3_u32x4 u32x4;

// v or zero, word by word (a select on the whole uint4 makes the compiler go through scratch memory)
__device__ __forceinline__ uint4 keep(uint4 v, bool ok) {
    return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
}

// XCD-aware workgroup order.  The dispatcher places workgroup b on XCD b % 8 and every XCD has its own L2, so workgroups
// that are neighbours in the launch order -- which here read the same activation rows (adjacent pixel tiles share their
// 3x3 halo, the output-channel tiles of a pixel tile share the whole input tile, the taps of a weight-gradient step read
// the same dY block and shifted copies of the same X block) -- each pull their own copy through a different L2.  The
// kernels are launched on a 1-D grid and take  v = xcd_order(blockIdx.x, gridDim.x)  as their position in that order:
// XCD x runs the x-th CONTIGUOUS chunk of it (bijective for any grid size).
__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// byte offset of piece j of row r in a [rows][8 x 16 B] LDS image
__device__ __forceinline__ int lds_piece(int r, int j) { return (r * 8 + (j ^ ((r >> 1) & 7))) * 16; }

// Epilogue modes.  The expand convolution of an MBConv block (1x1, 24..160 -> 144..960 channels) is so cheap next to the
// expanded tensor E0 it produces that E0 is never stored: the forward runs the convolution twice (STATS: the BatchNorm
// statistics of the rounded outputs, nothing written; BNACT: the same tiles again, act(scale * e0 + shift) written), and
// the BatchNorm backward recomputes the tiles it needs from the 6x smaller block input (BWD_REDUCE: the two sums of the
// BatchNorm backward against the incoming gradient dz; BWD_APPLY: the gradient at the convolution output).  Every mode
// rounds the accumulators to bf16 first: the values are those the stored tensor held.
enum { kModePlain = 0, kModeStats = 1, kModeBnAct = 2, kModeBwdReduce = 3, kModeBwdApply = 4 };

struct EpiArgs {
    const uint16_t* add;        // PLAIN mode, bf16 output: y = bf16(bf16(conv + bias) + add[m][co]) -- the gradient of a skip
    int ldadd;                  // connection added where the data gradient is written ([M][ldadd] bf16, 16-byte pieces)
    uint16_t* dx;               // BWD_APPLY of the whole-row streaming kernel: ALSO the data gradient of the 1x1 convolution,
    int lddx;                   // dx[m][ci] = sum_co dy[m][co] w[co][ci] (+ add[m][ci]), [M][lddx] bf16, from the tile in LDS
    const float* coef;          // [scale | shift | mean | invstd][Cout] (stp3_bn_finalize)
    const uint16_t* dz;         // gradient at the activation output, [M][ldz] bf16 (BWD_*)
    const float* gsums;         // [2][Cout]: sum g, sum g * xhat over all replicas (BWD_APPLY)
    float inv_count;
    int ldz, act;
};

template <int ACT>
__device__ __forceinline__ float epi_act(float v) {
    if (ACT == STP3_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == STP3_ACT_SWISH) return v * fast_sigmoid(v);
    return v;
}
template <int ACT>
__device__ __forceinline__ float epi_act_grad(float pre) {
    if (ACT == STP3_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
    if (ACT == STP3_ACT_SWISH) {
        const float sg = fast_sigmoid(pre);
        return sg * (1.f + pre * (1.f - sg));
    }
    return 1.f;
}

// BUF (the launcher sets it whenever the input and the weight tensor are each smaller than 2 GiB): the staging loads go through
// buffer resources -- a scalar base, one 32-bit byte offset per load, and the hardware's range check returning zeros for the
// offset kBufOob of a padding tap / a row beyond the matrix -- instead of selecting between a 64-bit address and the address of
// g_zero16 (which the compiler turned into exec-mask branches: 284 instructions per K step beside 16 MFMAs, 103 of them VALU
// and 115 SALU, `s_getpc` of g_zero16 ten times per step).
template <int BN, int MODE = kModePlain, int ACT = STP3_ACT_NONE, bool BUF = false>
__global__ __launch_bounds__(256) void conv2d_igemm_kernel(ConvDims d, int tiles_co, const uint16_t* __restrict__ x,
                                                           const uint16_t* __restrict__ w,
                                                           const float* __restrict__ bias, void* __restrict__ y,
                                                           float* __restrict__ stat_partial, EpiArgs ep = EpiArgs()) {
    constexpr int TP = BN == 128 ? 2 : 1;                 // 32-pixel MFMA tiles per wave
    constexpr int TC = 2;                                 // 32-channel MFMA tiles per wave
    constexpr int kStage = (kBM + BN) * kBK * 2;          // bytes of one staging buffer (pixel image + weight image)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BN == 128 ? (wave >> 1) : wave;        // which 32*TP pixels of the tile
    const int wn = BN == 128 ? (wave & 1) : 0;            // which 64 channels of the tile
    // position in the XCD-aware order: the output-channel tiles of one pixel tile are neighbours, then the next pixel tile
    const int v = xcd_order(blockIdx.x, gridDim.x);
    const int tile_m = v / tiles_co, tile_co = v - tile_m * tiles_co;
    const int m0 = tile_m * kBM;
    const int co0 = tile_co * BN;

    // ---- staging roles: piece column j (8 k), rows rr + 32 i --------------------------------------------
    const int j = tid & 7, rr = tid >> 3;
    const uint16_t* const zero = reinterpret_cast<const uint16_t*>(&g_zero16);
    const uint16_t* prow[4];                              // pixel rows: address of (hi0, wi0, channel 0), or `zero`
    uint32_t poff[4];                                     // BUF: the same as a byte offset from x (mod 2^32), kBufOob = no row
    int phi[4], pwi[4];                                   // ... and hi0, wi0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + rr + 32 * i;
        if (m < d.M) {
            const int wo = m % d.Wo;
            const int t = m / d.Wo;
            const int ho = t % d.Ho;
            const int n = t / d.Ho;
            phi[i] = ho * d.stride - d.pad_h;
            pwi[i] = wo * d.stride - d.pad_w;
            if (BUF) poff[i] = (uint32_t)(((n * d.H + phi[i]) * d.W + pwi[i]) * d.ldx) * 2u;
            else prow[i] = x + (ptrdiff_t)((n * d.H + phi[i]) * d.W + pwi[i]) * d.ldx;
        } else {
            phi[i] = -(1 << 28);                           // never inside the image
            pwi[i] = 0;
            if (BUF) poff[i] = kBufOob; else prow[i] = zero;
        }
    }
    const uint16_t* wrow[BN / 32];                        // weight rows: address of (co, k = 0), or `zero`
    uint32_t woff[BN / 32];                               // BUF: byte offset from w, kBufOob = no row
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
        const int co = co0 + rr + 32 * i;
        if (BUF) woff[i] = co < d.Cout ? (uint32_t)(co * d.Ktot) * 2u : kBufOob;
        else wrow[i] = co < d.Cout ? w + (size_t)co * d.Ktot : zero;
    }
    const stp3_buffer xbuf = make_buffer(x, BUF ? (uint32_t)((size_t)d.N * d.H * d.W * d.ldx * 2) : 0u);
    const stp3_buffer wbuf = make_buffer(w, BUF ? (uint32_t)((size_t)d.Cout * d.Ktot * 2) : 0u);
    // position of this thread's piece in K: k = step * 64 + j * 8 = tap * Cin + ci
    int tap = 0, ci = j * 8;
    while (ci >= d.Cin) { ci -= d.Cin; ++tap; }
    int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int taps = d.KH * d.KW;
    // pointwise layers (1x1, no padding: half of the launches of a step) never leave the image: no bounds tests.  (The
    // output may be LARGER than the padding implies -- the per-phase data gradients ask for it, ops._strided_dgrad -- and
    // then a single tap does fall off the bottom / right edge.)
    const bool pointwise = taps == 1 && d.pad_h == 0 && d.pad_w == 0 && (d.Ho - 1) * d.stride < d.H &&
                           (d.Wo - 1) * d.stride < d.W;

    u32x4 ra[4], rb[BN / 32];                              // first-class vectors: HIP's uint4 (a struct) ends up in scratch here
    auto load_step = [&]() {                               // the piece of the current (tap, ci) for every row of this thread
        const bool kvalid = tap < taps;
        const int dh = kh * d.dil_h, dw = kw * d.dil_w;
        const int toff = (dh * d.W + dw) * d.ldx + ci;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BUF) {
                // branch-free: the conditions are combined with `&` (every one is cheap and safe to evaluate), the offset of a
                // tap outside the image becomes kBufOob and the load returns zeros
                bool ok = kvalid & (poff[i] != kBufOob);
                if (!pointwise) {
                    const int hi = phi[i] + dh, wi = pwi[i] + dw;
                    ok = ok & ((unsigned)hi < (unsigned)d.H) & ((unsigned)wi < (unsigned)d.W);
                }
                ra[i] = buffer_load16(xbuf, ok ? poff[i] + (uint32_t)toff * 2u : kBufOob);
            } else {
                bool ok = kvalid && prow[i] != zero;
                if (!pointwise) {
                    const int hi = phi[i] + dh, wi = pwi[i] + dw;
                    ok = ok && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
                }
                ra[i] = *reinterpret_cast<const u32x4*>(ok ? prow[i] + toff : zero);
            }
        }
        const int kk = tap * d.Cin + ci;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            if (BUF) rb[i] = buffer_load16(wbuf, (kvalid & (woff[i] != kBufOob)) ? woff[i] + (uint32_t)kk * 2u : kBufOob);
            else rb[i] = *reinterpret_cast<const u32x4*>((kvalid && wrow[i] != zero) ? wrow[i] + kk : zero);
        }
        // advance to the next step: k += 64
        ci += kBK;
        while (ci >= d.Cin && tap < taps) {
            ci -= d.Cin;
            ++tap;
            if (++kw == d.KW) { kw = 0; ++kh; }
        }
    };
    auto store_step = [&](uint8_t* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(buf + lds_piece(rr + 32 * i, j)) = ra[i];
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<u32x4*>(buf + kBM * kBK * 2 + lds_piece(rr + 32 * i, j)) = rb[i];
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int steps = (d.Ktot + kBK - 1) / kBK;
    load_step();
    store_step(smem);
    __syncthreads();
    const int frow = lane & 31, fk = lane >> 5;            // fragment row / which 8-k half of a 16-k MFMA step
    for (int s = 0; s < steps; ++s) {
        uint8_t* cur = smem + (s & 1) * kStage;
        if (s + 1 < steps) load_step();
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            Frag fa[TC], fb[TP];
#pragma unroll
            for (int a = 0; a < TC; ++a)
                fa[a].u = *reinterpret_cast<const uint4*>(cur + kBM * kBK * 2 + lds_piece(wn * 64 + a * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int b = 0; b < TP; ++b)
                fb[b].u = *reinterpret_cast<const uint4*>(cur + lds_piece(wm * (32 * TP) + b * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int a = 0; a < TC; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0);
        }
        if (s + 1 < steps) store_step(smem + ((s + 1) & 1) * kStage);
        __syncthreads();
    }

    // ---- epilogue: D[row = channel 8*(r>>2) + 4*(lane>>5) + (r&3)][col = pixel lane&31] -> LDS [pixel][channel] -------
    // row stride BN + 8 elements: 16-byte aligned rows whose starts rotate through the banks
    const int ldt = BN + 8;
    if (d.out_f32) {
        float* tile = reinterpret_cast<float*>(smem);      // [128][BN + 8] float: 128 * 136 * 4 = 69 632 B <= 2 * kStage
#pragma unroll
        for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = wn * 64 + a * 32 + 8 * q + 4 * fk;
                    const int p = wm * (32 * TP) + b * 32 + frow;
                    float4 v = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
                    if (d.has_bias) {
                        v.x += co0 + c < d.Cout ? bias[co0 + c] : 0.f;
                        v.y += co0 + c + 1 < d.Cout ? bias[co0 + c + 1] : 0.f;
                        v.z += co0 + c + 2 < d.Cout ? bias[co0 + c + 2] : 0.f;
                        v.w += co0 + c + 3 < d.Cout ? bias[co0 + c + 3] : 0.f;
                    }
                    *reinterpret_cast<float4*>(tile + p * ldt + c) = v;
                }
        __syncthreads();
        float* yo = reinterpret_cast<float*>(y);
        for (int e = tid; e < kBM * (BN / 4); e += 256) {   // 16-byte pieces: 4 channels
            const int p = e / (BN / 4), c = (e - p * (BN / 4)) * 4;
            const int m = m0 + p, co = co0 + c;
            if (m >= d.M || co >= d.Cout) continue;
            const float4 v = *reinterpret_cast<const float4*>(tile + p * ldt + c);
            float* dst = yo + (size_t)m * d.ldy + co;
            if (co + 3 < d.Cout && (d.ldy & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                dst[0] = v.x;
                if (co + 1 < d.Cout) dst[1] = v.y;
                if (co + 2 < d.Cout) dst[2] = v.z;
                if (co + 3 < d.Cout) dst[3] = v.w;
            }
        }
        return;
    }
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem);    // [128][BN + 8] bf16
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = wn * 64 + a * 32 + 8 * q + 4 * fk;
                const int p = wm * (32 * TP) + b * 32 + frow;
                float v[4] = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (d.has_bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += co0 + c + r < d.Cout ? bias[co0 + c + r] : 0.f;
                }
                *reinterpret_cast<uint2*>(tile + p * ldt + c) =
                    make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
            }
    __syncthreads();
    uint16_t* yo = reinterpret_cast<uint16_t*>(y);
    if (MODE == kModePlain) {
        for (int e = tid; e < kBM * (BN / 8); e += 256) {       // 16-byte pieces: 8 channels
            const int p = e / (BN / 8), c = (e - p * (BN / 8)) * 8;
            const int m = m0 + p, co = co0 + c;
            if (m >= d.M || co >= d.Cout) continue;
            uint4 v = *reinterpret_cast<const uint4*>(tile + p * ldt + c);
            uint16_t* dst = yo + (size_t)m * d.ldy + co;
            if (co + 7 < d.Cout && (d.ldy & 7) == 0) {
                if (ep.add) {                                  // (the host passes `add` only for whole 16-byte pieces)
                    const uint4 a = *reinterpret_cast<const uint4*>(ep.add + (size_t)m * ep.ldadd + co);
                    const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, aw[4] = {a.x, a.y, a.z, a.w};
                    uint32_t ow[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ow[r] = pack_bf16(__uint_as_float(vw[r] << 16) + __uint_as_float(aw[r] << 16),
                                          __uint_as_float(vw[r] & 0xffff0000u) + __uint_as_float(aw[r] & 0xffff0000u));
                    v = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                }
                *reinterpret_cast<uint4*>(dst) = v;
            } else {
                const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (co + r < d.Cout) dst[r] = (uint16_t)(wds[r >> 1] >> (16 * (r & 1)));
            }
        }
    }
    if (MODE == kModeBnAct || MODE == kModeBwdApply) {
        // a thread's pieces all hold the same 8 channels (256 threads are a multiple of the BN / 8 pieces of a pixel):
        // the per-channel constants live in registers.  Cout, ldy (and ldz) are multiples of 8 here (checked by the host).
        const int c = (tid % (BN / 8)) * 8, co = co0 + c;
        if (co < d.Cout) {
            float cs[8], ct[8], a2[8], a3[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                cs[r] = ep.coef[co + r];
                ct[r] = ep.coef[d.Cout + co + r];
                if (MODE == kModeBwdApply) {
                    const float mu = ep.coef[2 * d.Cout + co + r], is = ep.coef[3 * d.Cout + co + r];
                    const float k0 = ep.gsums[co + r] * ep.inv_count, k1 = ep.gsums[d.Cout + co + r] * ep.inv_count;
                    a2[r] = -(cs[r] * is) * k1;                      // the constants of bn_apply_bwd_kernel (stp3_bnact.hip)
                    a3[r] = -cs[r] * k0 - a2[r] * mu;
                }
            }
            for (int p = tid / (BN / 8); p < kBM; p += 256 / (BN / 8)) {
                const int m = m0 + p;
                if (m >= d.M) break;
                const uint4 v = *reinterpret_cast<const uint4*>(tile + p * ldt + c);
                const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
                float e0[8], out[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e0[2 * r] = __uint_as_float(wds[r] << 16);
                    e0[2 * r + 1] = __uint_as_float(wds[r] & 0xffff0000u);
                }
                if (MODE == kModeBnAct) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) out[r] = epi_act<ACT>(fmaf(e0[r], cs[r], ct[r]));
                } else {
                    const uint4 g = *reinterpret_cast<const uint4*>(ep.dz + (size_t)m * ep.ldz + co);
                    const uint32_t gds[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float dzv = __uint_as_float((r & 1) ? (gds[r >> 1] & 0xffff0000u) : (gds[r >> 1] << 16));
                        const float gg = dzv * epi_act_grad<ACT>(fmaf(e0[r], cs[r], ct[r]));
                        out[r] = fmaf(cs[r], gg, fmaf(a2[r], e0[r], a3[r]));
                    }
                }
                *reinterpret_cast<uint4*>(yo + (size_t)m * d.ldy + co) =
                    make_uint4(pack_bf16(out[0], out[1]), pack_bf16(out[2], out[3]), pack_bf16(out[4], out[5]),
                               pack_bf16(out[6], out[7]));
            }
        }
    }
    if (MODE == kModeBwdReduce) {
        // the gradient tile dz[128][BN] goes through LDS too (16-byte loads), behind the output tile; then thread
        // (channel, pixel part) adds its pixels: sum g and sum g * xhat, one partial row per workgroup
        uint16_t* gt = tile + kBM * ldt;
        for (int e = tid; e < kBM * (BN / 8); e += 256) {
            const int p = e / (BN / 8), c = (e - p * (BN / 8)) * 8;
            const int m = m0 + p, co = co0 + c;
            uint4 g = make_uint4(0u, 0u, 0u, 0u);
            if (m < d.M && co < d.Cout) g = *reinterpret_cast<const uint4*>(ep.dz + (size_t)m * ep.ldz + co);
            *reinterpret_cast<uint4*>(gt + p * ldt + c) = g;
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + 2 * kBM * ldt * 2);
        const int c = tid % BN, part = tid / BN;
        constexpr int kParts = 256 / BN, kRows = kBM / kParts;
        float s1 = 0.f, s2 = 0.f;
        if (co0 + c < d.Cout) {
            const float cs = ep.coef[co0 + c], ct = ep.coef[d.Cout + co0 + c];
            const float mu = ep.coef[2 * d.Cout + co0 + c], is = ep.coef[3 * d.Cout + co0 + c];
            for (int p = part * kRows; p < (part + 1) * kRows; ++p) {
                if (m0 + p < d.M) {
                    const float e0 = bf2f(tile[p * ldt + c]);
                    const float gg = bf2f(gt[p * ldt + c]) * epi_act_grad<ACT>(fmaf(e0, cs, ct));
                    s1 += gg;
                    s2 = fmaf(gg, (e0 - mu) * is, s2);
                }
            }
        }
        red[(part * 2) * BN + c] = s1;
        red[(part * 2 + 1) * BN + c] = s2;
        __syncthreads();
        if (tid < 2 * BN) {
            const int k = tid / BN, cc = tid - k * BN;
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < kParts; ++q) t += red[(q * 2 + k) * BN + cc];
            if (co0 + cc < d.Cout) stat_partial[((size_t)tile_m * 2 + k) * d.Cout + co0 + cc] = t;
        }
    }
    if ((MODE == kModePlain || MODE == kModeStats) && stat_partial) {
        // BatchNorm statistics of the rounded outputs: thread (channel c, pixel half) adds 64 pixels; rows beyond M
        // hold the bias only and are skipped
        float* red = reinterpret_cast<float*>(smem + kBM * ldt * 2);        // behind the tile: 2 * 2 * BN floats
        const int c = tid % BN, part = tid / BN;                            // BN = 128: 2 parts; BN = 64: 4 parts
        constexpr int kParts = 256 / BN, kRows = kBM / kParts;
        float s1 = 0.f, s2 = 0.f;
        for (int p = part * kRows; p < (part + 1) * kRows; ++p) {
            if (m0 + p < d.M) {
                const float v = bf2f(tile[p * ldt + c]);
                s1 += v;
                s2 += v * v;
            }
        }
        red[(part * 2) * BN + c] = s1;
        red[(part * 2 + 1) * BN + c] = s2;
        __syncthreads();
        if (tid < 2 * BN) {
            const int k = tid / BN, cc = tid - k * BN;
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < kParts; ++q) t += red[(q * 2 + k) * BN + cc];
            if (co0 + cc < d.Cout) stat_partial[((size_t)tile_m * 2 + k) * d.Cout + co0 + cc] = t;
        }
    }
}

// ---- the epilogue of the streaming kernels on 8 channels of one pixel, TWO values per instruction ---------------------------
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the BatchNorm + swish passes of these kernels are co-limited by their vector
// instructions at 2-3 waves per SIMD; the packed forms halve everything but the exponential and the reciprocal.  Same
// operations in the same order as the scalar statements of the tiled kernel: same bits.)
struct Chan8 {
    stp3_f32x2 cs[4], ct[4], mu[4], is[4], a2[4], a3[4];
};

template <int MODE>
__device__ __forceinline__ void load_chan8(Chan8& k, const EpiArgs& ep, int Cout, int c, bool ok) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            k.cs[r][h] = k.ct[r][h] = k.mu[r][h] = k.is[r][h] = k.a2[r][h] = k.a3[r][h] = 0.f;
            if (MODE >= kModeBnAct && ok) {
                const int cc = c + 2 * r + h;
                k.cs[r][h] = ep.coef[cc];
                k.ct[r][h] = ep.coef[Cout + cc];
                k.mu[r][h] = ep.coef[2 * Cout + cc];
                k.is[r][h] = ep.coef[3 * Cout + cc];
                if (MODE == kModeBwdApply) {
                    const float k0 = ep.gsums[cc] * ep.inv_count, k1 = ep.gsums[Cout + cc] * ep.inv_count;
                    k.a2[r][h] = -(k.cs[r][h] * k.is[r][h]) * k1;
                    k.a3[r][h] = -k.cs[r][h] * k0 - k.a2[r][h] * k.mu[r][h];
                }
            }
        }
}

// wds: the 8 rounded outputs (bf16 pairs); gds: the 8 incoming gradients (BWD_* modes); s1 / s2: the running sums (PLAIN /
// STATS: sum, sum of squares; BWD_REDUCE: sum g, sum g * xhat); ow: the 8 results as bf16 pairs (BNACT / BWD_APPLY)
template <int MODE>
__device__ __forceinline__ void epi8(const uint32_t (&wds)[4], const uint32_t (&gds)[4], const Chan8& k, int act,
                                     stp3_f32x2 (&s1)[4], stp3_f32x2 (&s2)[4], uint32_t (&ow)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const stp3_f32x2 e0 = {__uint_as_float(wds[r] << 16), __uint_as_float(wds[r] & 0xffff0000u)};
        if (MODE == kModePlain || MODE == kModeStats) {
            s1[r] = s1[r] + e0;
            s2[r] = pk_fma(e0, e0, s2[r]);
        } else {
            const stp3_f32x2 pre = pk_fma(e0, k.cs[r], k.ct[r]);
            if (MODE == kModeBnAct) {
                stp3_f32x2 out = pre;
                if (act == STP3_ACT_SWISH) out = pre * pk_sigmoid(pre);
                else if (act == STP3_ACT_RELU) out = stp3_f32x2{fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f)};
                ow[r] = pack_bf16(out.x, out.y);
            } else {
                const stp3_f32x2 dz = {__uint_as_float(gds[r] << 16), __uint_as_float(gds[r] & 0xffff0000u)};
                stp3_f32x2 der = {1.f, 1.f};
                if (act == STP3_ACT_SWISH) {
                    const stp3_f32x2 sg = pk_sigmoid(pre);
                    der = sg * (1.f + pre * (1.f - sg));
                } else if (act == STP3_ACT_RELU) {
                    der = stp3_f32x2{pre.x > 0.f ? 1.f : 0.f, pre.y > 0.f ? 1.f : 0.f};
                }
                const stp3_f32x2 gg = dz * der;
                if (MODE == kModeBwdReduce) {
                    s1[r] = s1[r] + gg;
                    s2[r] = pk_fma(gg, (e0 - k.mu[r]) * k.is[r], s2[r]);
                } else {
                    const stp3_f32x2 out = pk_fma(k.cs[r], gg, pk_fma(k.a2[r], e0, k.a3[r]));
                    ow[r] = pack_bf16(out.x, out.y);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1) convolutions with a SHORT contraction (Cin <= 128): the expand convolutions of the MBConv blocks
// and the data gradients of their project convolutions -- 24..112 channels in, 6x as many out, activations of 40..560 MB.
// The tiled kernel above spends its time per WORKGROUP on them (one K-step of matrix work between an address prologue, two
// LDS round trips and three barriers: 24 -> 144 @112x240x72 ran at 2.5 TB/s of its own traffic): these layers are
// streaming kernels with a small matrix product inside, and are written as such -- persistent waves, operand fragments
// straight from global memory into registers (the next tile's in flight), no workgroup barrier in the loop, ONE partial row
// of statistics per workgroup / pixel group -- in two forms, chosen by what the STORES need:
//
// pointwise_rows_kernel (Cout = 144 / 192, Cin <= 32: the two layers whose outputs -- 532 and 177 MiB at B = 4 -- do not fit
// the 256 MB infinity cache).  scripts/probe_stores.hip: a 532-MiB tensor of 288-byte pixel rows is written at 5.6 TB/s when
// every wave writes WHOLE pixel rows, at 3.1 TB/s when it writes the 128 bytes of a 64-channel block (what a channel-tiled
// kernel does: the rest of each 128-byte line arrives from another wave, later) and at 2.1 TB/s in 32-byte pieces.  So a
// wave computes ALL output channels of its 32 pixels (5 / 6 MFMA tiles, the weight image in LDS once per workgroup), turns
// the accumulators into pixel rows through a wave-private LDS tile and stores them as one contiguous 9 / 12 KB run; a lane
// keeps the same 8 channels in every pass (64 / PP whole rows per store instruction), so their BatchNorm constants and
// partial sums live in registers.
template <int PP, int KMAX, int MODE>
__global__ __launch_bounds__(256, 2) void pointwise_rows_kernel(ConvDims d, int ksteps, int tiles_co, const uint16_t* __restrict__ x,
                                                             const uint16_t* __restrict__ w, uint16_t* __restrict__ y,
                                                             float* __restrict__ stat_partial, EpiArgs ep) {
    constexpr int NT = (PP + 3) / 4;                       // 32-channel MFMA tiles
    constexpr int RPI = 64 / PP;                           // whole pixel rows per store instruction
    constexpr int NIT = (32 + RPI - 1) / RPI;              // store instructions per 32-pixel tile
    constexpr int LDT = NT * 32 + 8;                       // row stride of the transposition tile (elements)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup = (pixel group, channel tile of PP * 8 channels): ONE tile for the 144 / 192-channel layers (whole rows);
    // several for Cout = k * 64 / k * 128 (whole 128-byte lines); tile fastest within an XCD's contiguous chunk of the order
    const int v_order = xcd_order(blockIdx.x, gridDim.x);
    const int group = v_order / tiles_co, ngroups = gridDim.x / tiles_co;
    const int co0 = (v_order - group * tiles_co) * (PP * 8);
    const int LDW = ksteps * 16 + 8;                       // row stride of the weight image (elements)
    uint16_t* wimg = reinterpret_cast<uint16_t*>(smem);    // [NT * 32][LDW]
    uint16_t* tile = wimg + NT * 32 * LDW + wave * (32 * LDT);
    float* red = reinterpret_cast<float*>(wimg + NT * 32 * LDW);   // [4 waves][64 lanes][16], over the tiles at the end
    // BWD_APPLY with ep.dx: the TRANSPOSED weight image [32 input channels][LDT] behind the four wave tiles -- the second
    // matrix product of the pass, dx^T[ci][pixel] = sum_co w[co][ci] dy[pixel][co], reads dy from the wave's tile
    uint16_t* wtimg = wimg + NT * 32 * LDW + 4 * (32 * LDT);
    const bool with_dx = MODE == kModeBwdApply && ep.dx != nullptr;
    if (with_dx) {
        for (int e = tid; e < 32 * NT * 32; e += 256) {
            const int k = e / (NT * 32), c = e - k * (NT * 32);
            wtimg[k * LDT + c] = (c < PP * 8 && k < d.Cin) ? w[(size_t)(co0 + c) * d.Cin + k] : (uint16_t)0;
        }
    }
    for (int e = tid; e < NT * 32 * ksteps * 2; e += 256) {
        const int r = e / (ksteps * 2), k0 = (e - r * (ksteps * 2)) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < PP * 8 && k0 < d.Cin) v = *reinterpret_cast<const u32x4*>(w + (size_t)(co0 + r) * d.Cin + k0);
        *reinterpret_cast<u32x4*>(wimg + r * LDW + k0) = v;
    }
    __syncthreads();
    const int px = lane & 31, half = lane >> 5;
    const int cp = lane % PP, prow = lane / PP;
    const bool lane_ok = prow < RPI;
    const int cl = cp * 8, cch = co0 + cl;                 // this lane's 8 channels in the store passes: in the tile, in the tensor
    Chan8 kc;
    load_chan8<MODE>(kc, ep, d.Cout, cch, true);
    stp3_f32x2 s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[r] = s2[r] = stp3_f32x2{0.f, 0.f};
    const uint16_t* const zero = reinterpret_cast<const uint16_t*>(&g_zero16);
    u32x4 cur[KMAX], nxt[KMAX];
    const int ntiles = (d.M + 31) / 32;
    const int stride_t = ngroups * 4;
    auto load_tile = [&](int t, u32x4 (&f)[KMAX]) {
        const int m = t * 32 + px;
        const uint16_t* row = (t < ntiles && m < d.M) ? x + (size_t)m * d.ldx : nullptr;
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) {
            const int k0 = ks * 16 + half * 8;
            if (ks < ksteps) f[ks] = *reinterpret_cast<const u32x4*>((row && k0 < d.Cin) ? row + k0 : zero);
        }
    };
    int t = group * 4 + wave;
    load_tile(t, cur);
    for (; t < ntiles; t += stride_t) {
        load_tile(t + stride_t, nxt);
        const int mbase = t * 32;
        f32x16 acc[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) {
            if (ks < ksteps) {
                Frag fb;
                fb.u = make_uint4(cur[ks][0], cur[ks][1], cur[ks][2], cur[ks][3]);
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    Frag fa;
                    fa.u = *reinterpret_cast<const uint4*>(wimg + (a * 32 + px) * LDW + ks * 16 + half * 8);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[a], 0, 0, 0);
                }
            }
        }
        // D[row = channel 8*(r>>2) + 4*half + (r&3)][col = pixel px] -> the wave's tile [pixel][channel], rounded to bf16
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint2*>(tile + px * LDT + a * 32 + 8 * q + 4 * half) =
                    make_uint2(pack_bf16(acc[a][4 * q], acc[a][4 * q + 1]), pack_bf16(acc[a][4 * q + 2], acc[a][4 * q + 3]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- store passes: RPI whole pixel rows per instruction, this lane's 8 channels of row prow + i * RPI
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int p = prow + i * RPI;
            const int m = mbase + p;
            if (lane_ok && p < 32 && m < d.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(tile + p * LDT + cl);
                if (MODE == kModePlain) *reinterpret_cast<uint4*>(y + (size_t)m * d.ldy + cch) = v;
                if (MODE != kModePlain || stat_partial) {
                    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
                    uint32_t gds[4] = {0u, 0u, 0u, 0u}, ow[4];
                    if (MODE >= kModeBwdReduce) {
                        const uint4 g = *reinterpret_cast<const uint4*>(ep.dz + (size_t)m * ep.ldz + cch);
                        gds[0] = g.x; gds[1] = g.y; gds[2] = g.z; gds[3] = g.w;
                    }
                    epi8<MODE>(wds, gds, kc, ep.act, s1, s2, ow);
                    if (MODE == kModeBnAct || MODE == kModeBwdApply)
                        *reinterpret_cast<uint4*>(y + (size_t)m * d.ldy + cch) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    if (with_dx)                            // dy replaces the recomputed output in the wave's tile
                        *reinterpret_cast<uint4*>(tile + p * LDT + cl) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                }
            }
        }
        if (with_dx) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x16 dxa;
#pragma unroll
            for (int r = 0; r < 16; ++r) dxa[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NT * 2; ++ks) {
                Frag fa, fb;
                fa.u = *reinterpret_cast<const uint4*>(wtimg + px * LDT + ks * 16 + half * 8);
                fb.u = *reinterpret_cast<const uint4*>(tile + px * LDT + ks * 16 + half * 8);
                dxa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, dxa, 0, 0, 0);
            }
            // D[row = input channel 8 q + 4 half + j][col = pixel px]: four 8-byte pieces of the pixel's dx row per lane
            const int m = mbase + px;
            if (m < d.M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = 8 * q + 4 * half;
                    if (ci < d.Cin) {
                        uint32_t lo = pack_bf16(dxa[4 * q], dxa[4 * q + 1]), hi = pack_bf16(dxa[4 * q + 2], dxa[4 * q + 3]);
                        if (ep.add) {                      // + the gradient of the block's skip (added to the ROUNDED product)
                            const uint2 a = *reinterpret_cast<const uint2*>(ep.add + (size_t)m * ep.ldadd + ci);
                            lo = pack_bf16(__uint_as_float(lo << 16) + __uint_as_float(a.x << 16),
                                           __uint_as_float(lo & 0xffff0000u) + __uint_as_float(a.x & 0xffff0000u));
                            hi = pack_bf16(__uint_as_float(hi << 16) + __uint_as_float(a.y << 16),
                                           __uint_as_float(hi & 0xffff0000u) + __uint_as_float(a.y & 0xffff0000u));
                        }
                        *reinterpret_cast<uint2*>(ep.dx + (size_t)m * ep.lddx + ci) = make_uint2(lo, hi);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                   // the tile is rewritten by the next iteration
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) cur[ks] = nxt[ks];
    }
    // ---- one partial row per workgroup: the lanes' sums meet in LDS, added in a fixed order (wave, pixel-row group)
    if (stat_partial && (MODE == kModePlain || MODE == kModeStats || MODE == kModeBwdReduce)) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            red[(wave * 64 + lane) * 16 + r] = s1[r >> 1][r & 1];
            red[(wave * 64 + lane) * 16 + 8 + r] = s2[r >> 1][r & 1];
        }
        __syncthreads();
        for (int e = tid; e < 2 * PP * 8; e += 256) {
            const int k = e / (PP * 8), c = e - k * (PP * 8);   // which sum, which channel of the tile
            const int pc = c >> 3, r = c & 7;
            float tot = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int g = 0; g < RPI; ++g) tot += red[(wv * 64 + g * PP + pc) * 16 + k * 8 + r];
            stat_partial[((size_t)group * 2 + k) * d.Cout + co0 + c] = tot;
        }
    }
}

// pointwise_direct_kernel (every other qualifying layer: outputs that stay in the infinity cache, where the store pattern
// costs little) has no LDS transposition: the order of the weight rows in the A operand is free, so the 32 output
// channels of an MFMA tile are permuted such that the 16 accumulators of a lane ARE two 16-byte channel pieces of its pixel:
//     MFMA row 8q + 4h + j (q = 0..3, h = lane >> 5, j = 0..3)  <->  channel 16 (q >> 1) + 8 h + 4 (q & 1) + j
//     => acc[8s .. 8s + 7] of lane (pixel, h) = channels 16 s + 8 h .. + 7            (s = 0, 1)
// A WAVE owns one 32-channel block for the whole kernel -- its weight fragments, per-channel constants and partial sums live
// in registers -- and walks over 32-pixel tiles: operand fragments straight from global memory (next tile in flight), 1..8
// MFMAs, the epilogue on the accumulators, two 16-byte stores per lane (the lanes l and l + 32 of a pixel write adjacent
// pieces; the L2 merges a line's pieces from the waves of neighbouring channel blocks, which run on the same XCD).  No LDS,
// no barrier in the loop, >= 4 waves per SIMD for the thin layers.
template <int KMAX, int MODE>
__global__ __launch_bounds__(256) void pointwise_direct_kernel(ConvDims d, int ksteps, int cblocks, int ngroups,
                                                               const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                               uint16_t* __restrict__ y, float* __restrict__ stat_partial,
                                                               EpiArgs ep) {
    constexpr bool kSums = MODE == kModePlain || MODE == kModeStats || MODE == kModeBwdReduce;
    __shared__ float red[kSums ? 4 * 64 * 33 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = xcd_order(blockIdx.x, gridDim.x) * 4 + wave;      // consecutive slots share an XCD (and its L2)
    const int group = slot / cblocks, cb = slot - group * cblocks;
    const bool live = group < ngroups;
    const int px = lane & 31, h = lane >> 5;
    const uint16_t* const zero = reinterpret_cast<const uint16_t*>(&g_zero16);
    // ---- weight fragments: MFMA row px holds channel cb * 32 + perm(px)
    Frag wf[KMAX];
    {
        const int q = px >> 3, hh = (px >> 2) & 1, j = px & 3;
        const int co = cb * 32 + 16 * (q >> 1) + 8 * hh + 4 * (q & 1) + j;
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) {
            const int k0 = ks * 16 + h * 8;
            wf[ks].u = *reinterpret_cast<const uint4*>((live && ks < ksteps && co < d.Cout && k0 < d.Cin) ? w + (size_t)co * d.Cin + k0 : zero);
        }
    }
    // ---- this lane's two channel pieces and their constants
    int ch[2];
    bool ok[2];
    Chan8 kc[2];
    stp3_f32x2 s1[2][4], s2[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ch[s] = cb * 32 + 16 * s + 8 * h;
        ok[s] = live && ch[s] < d.Cout;
        load_chan8<MODE>(kc[s], ep, d.Cout, ch[s], ok[s]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[s][r] = s2[s][r] = stp3_f32x2{0.f, 0.f};
    }
    const int ntiles = (d.M + 31) / 32;
    u32x4 cur[KMAX], nxt[KMAX];
    auto load_tile = [&](int t, u32x4 (&f)[KMAX]) {
        const int m = t * 32 + px;
        const uint16_t* row = (live && t < ntiles && m < d.M) ? x + (size_t)m * d.ldx : nullptr;
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) {
            if (ks < ksteps) {
                const int k0 = ks * 16 + h * 8;
                f[ks] = *reinterpret_cast<const u32x4*>((row && k0 < d.Cin) ? row + k0 : zero);
            }
        }
    };
    int t = live ? group : ntiles;
    load_tile(t, cur);
    for (; t < ntiles; t += ngroups) {
        load_tile(t + ngroups, nxt);
        const int m = t * 32 + px;
        const bool row_ok = m < d.M;
        uint4 g[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
        if (MODE >= kModeBwdReduce) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (row_ok && ok[s]) g[s] = *reinterpret_cast<const uint4*>(ep.dz + (size_t)m * ep.ldz + ch[s]);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) {
            if (ks < ksteps) {
                Frag fb;
                fb.u = make_uint4(cur[ks][0], cur[ks][1], cur[ks][2], cur[ks][3]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks].v, fb.v, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the rounded outputs (what the stored tensor holds) and their float32 values
            const uint32_t wds[4] = {pack_bf16(acc[8 * s], acc[8 * s + 1]), pack_bf16(acc[8 * s + 2], acc[8 * s + 3]),
                                     pack_bf16(acc[8 * s + 4], acc[8 * s + 5]), pack_bf16(acc[8 * s + 6], acc[8 * s + 7])};
            if (!(row_ok && ok[s])) continue;
            if (MODE == kModePlain) {
                *reinterpret_cast<uint4*>(y + (size_t)m * d.ldy + ch[s]) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                if (!stat_partial) continue;
            }
            const uint32_t gds[4] = {g[s].x, g[s].y, g[s].z, g[s].w};
            uint32_t ow[4];
            epi8<MODE>(wds, gds, kc[s], ep.act, s1[s], s2[s], ow);
            if (MODE == kModeBnAct || MODE == kModeBwdApply)
                *reinterpret_cast<uint4*>(y + (size_t)m * d.ldy + ch[s]) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
#pragma unroll
        for (int ks = 0; ks < KMAX; ++ks) cur[ks] = nxt[ks];
    }
    // ---- the lanes of a half-wave hold partial sums of the same 16 channels: added over the 32 pixels lanes through LDS,
    // in lane order; one partial row per pixel group
    if (kSums) {
        if (!stat_partial) return;                                      // (uniform: a plain convolution without statistics)
        float* mine = red + (wave * 64 + lane) * 33;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                mine[s * 8 + r] = s1[s][r >> 1][r & 1];
                mine[16 + s * 8 + r] = s2[s][r >> 1][r & 1];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // lane (h, idx): idx = k * 16 + s * 8 + r  ->  channel cb * 32 + 16 s + 8 h + r of sum k
        const int idx = lane & 31;
        float tot = 0.f;
        for (int p = 0; p < 32; ++p) tot += red[(wave * 64 + h * 32 + p) * 33 + idx];
        const int k = idx >> 4, sp = (idx >> 3) & 1, r = idx & 7;
        const int c = cb * 32 + 16 * sp + 8 * h + r;
        if (live && c < d.Cout) stat_partial[((size_t)group * 2 + k) * d.Cout + c] = tot;
    }
}

// Column sums of a [parts][width] float32 matrix, deterministic, in two coalesced levels: workgroup (x, y) adds rows
// [256 y, 256 y + 256) of columns [64 x, 64 x + 64) (4 row lanes x 64 columns, double accumulation) into
// out[y][...]; a second launch with the level-1 result as input finishes (parts <= 256: one level).
__global__ __launch_bounds__(256) void colsum_kernel(int parts, int width, const float* __restrict__ partial,
                                                     float* __restrict__ out) {
    __shared__ double red[256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * 256, r1 = min(r0 + 256, parts);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (col < width) {
        int p = r0 + rl;
        for (; p + 12 < r1; p += 16) {                     // four independent loads in flight
            const float a = partial[(size_t)p * width + col], b = partial[(size_t)(p + 4) * width + col];
            const float c = partial[(size_t)(p + 8) * width + col], e = partial[(size_t)(p + 12) * width + col];
            s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)e;
        }
        for (; p < r1; p += 4) s0 += (double)partial[(size_t)p * width + col];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && col < width)
        out[(size_t)blockIdx.y * width + col] = (float)(red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl]);
}

// sums[width] = column sums of partial[parts][width]; `partial` must have room for ceil(parts / 256) more rows behind
// its `parts` rows (the level-1 result)
inline void launch_colsum(hipStream_t s, int parts, int width, float* partial, float* sums) {
    const int wx = (width + 63) / 64;
    if (parts <= 256) {
        hipLaunchKernelGGL(colsum_kernel, dim3(wx, 1), dim3(256), 0, s, parts, width, (const float*)partial, sums);
        return;
    }
    const int l1 = (parts + 255) / 256;
    float* mid = partial + (size_t)parts * width;
    hipLaunchKernelGGL(colsum_kernel, dim3(wx, l1), dim3(256), 0, s, parts, width, (const float*)partial, mid);
    // l1 <= 256 for parts <= 65536 (8.4 M pixels); beyond that the second level loops over more rows per workgroup
    hipLaunchKernelGGL(colsum_kernel, dim3(wx, 1), dim3(256), 0, s, l1 > 256 ? 256 : l1, width, (const float*)mid, sums);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[co][tap][ci] = sum_m dY[m][co] * X[pixel(m) + tap][ci]
// ------------------------------------------------------------------------------------------------
// The contraction runs over PIXELS, the slow dimension of both NHWC operands, while an MFMA fragment wants 8
// consecutive k per lane.  gfx950 has the instruction for exactly this: the staging copies 16-byte pieces (8 channels
// of a pixel) global -> LDS as they lie, into [64 pixels][T channels] images, and the fragments are read with the LDS
// TRANSPOSE read (ds_read_b64_tr_b16, stp3_cdna.h): two reads give a lane its 8 pixels of one channel.  dY^T is
// operand A (rows = output channels), X^T operand B: an accumulator lane holds dW[co ..][ci = lane & 31], i.e.
// coalesced float32 rows of the result.  Workgroup tile TCO x TCI (64 or 128 each) of one tap; the pixels are split
// over gridDim.z workgroups (float32 partial sums, reduced deterministically afterwards).
// (Rounds 1-2 transposed 8 x 8 blocks in registers, 32 x v_perm_b32 per thread and step, and carried (n, ho, wo) per
// loaded piece: 332 VALU instructions beside 4 MFMAs per step, profiles/r03e_wgrad_pmc.json -- VALU-bound.  Now the
// pixel arithmetic is done ONCE per pixel and step -- lane l of every wave owns pixel 64 * step + l: coordinates
// advanced incrementally, bounds test, input pixel index -- and a loading thread fetches the index of its pixel from
// that lane with one ds_bpermute.)
// Tap folding (tap_fold > 0; layers with Cin == 8, i.e. the 3-channel stem): one channel block holds ALL input
// channels, so the TCI / 8 piece slots of an X row carry tap_fold = TCI / 8 different TAPS instead -- the tile's
// columns are (tap, ci) pairs, contiguous in dW -- and dY, the big operand (48 channels x 1.9 M pixels for the stem),
// is read once per tap group instead of once per tap.  The bounds test then depends on the loading thread's tap: the
// pixel lane hands over the un-shifted pixel index and the packed (row, column) instead.

// byte offset of the 16-byte piece p of pixel row m in a [64][T] bf16 image.  The pieces of a row are permuted so that
// the four rows x 64 bytes one half-wave touches in a transpose read fall into 64 different banks (128-byte rows: rows
// m and m + 2 share their banks -> swap the halves of the odd row pair; 256-byte rows: all rows do -> rotate by m & 3)
template <int T>
__device__ __forceinline__ int wg_piece(int m, int p) {
    const int sw = T == 64 ? ((m >> 1) & 1) << 2 : (m & 3) << 2;
    return m * (T * 2) + ((p ^ sw) << 4);
}

union FragTr {
    stp3_s16x4 h[2];
    bf16x8 v;
};

// BUF: the staging loads through buffer resources (see conv2d_igemm_kernel): both operands smaller than 2 GiB
template <int TCO, int TCI, bool BUF = false>
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(ConvDims d, int tiles_ci, int ksteps_per_block, int tap_fold,
                                                           int xcd_chunks, const uint16_t* __restrict__ dy,
                                                           const uint16_t* __restrict__ x,
                                                           float* __restrict__ partial) {
    constexpr int TA = TCO / 64, TB = TCI / 64;            // 32-row MFMA tiles per wave (2 x 2 waves)
    constexpr int kImgA = kBK * TCO * 2;                   // bytes of the dY image; the X image lies behind it
    constexpr int kStage = kBK * (TCO + TCI) * 2;
    constexpr int PA = TCO / 8, PB = TCI / 8;              // 16-byte pieces per pixel row
    constexpr int NA = TCO / 32, NB = TCI / 32;            // pieces per thread and step
    constexpr int RA = 256 / PA, RB = 256 / PB;            // pixel rows between two pieces of a thread
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;               // co half / ci half of the tile
    // Workgroup order.  As dispatched (tile, tap, split) the nine taps of a pixel split land on nine different XCDs and each
    // pulls its own copy of the dY / X rows through its own L2 (PMC: 7 % hit rate, 4.3x the operands fetched).  With
    // `xcd_chunks` (set by the launcher) XCD x runs a contiguous chunk of that order instead -- the taps of a split share
    // one L2: 3x3 64 -> 64 @200x200x12 173 -> 108 us, 3x3 d12 64 -> 128 246 -> 157 us (profiles/r03r_time_conv_wgrad_xcd.txt).  (On the round-2
    // kernel, which was bound by its instruction count, the same order was SLOWER; it also is for 49 taps.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd_chunks) {
        const int lin = xcd_order(bx + (int)gridDim.x * (by + (int)gridDim.y * bz), (int)(gridDim.x * gridDim.y * gridDim.z));
        bx = lin % (int)gridDim.x;
        by = (lin / (int)gridDim.x) % (int)gridDim.y;
        bz = lin / (int)(gridDim.x * gridDim.y);
    }
    const int tco = bx / tiles_ci, tci = bx - tco * tiles_ci;
    const int tap = by;
    const int split = bz;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int co0 = tco * TCO, ci0 = tci * TCI;
    const int total_steps = (d.M + kBK - 1) / kBK;
    const int step0 = split * ksteps_per_block;
    const int step1 = min(step0 + ksteps_per_block, total_steps);
    const uint16_t* const zero = reinterpret_cast<const uint16_t*>(&g_zero16);

    // ---- loading roles (fixed for the kernel): piece pa of the dY rows ra0 + RA * i, piece pb of the X rows rb0 + RB * i.
    // A thread whose channels lie beyond the tensor reads the zero page whatever the pixel: base = zero page, pitch = 0.
    const int pa = tid % PA, ra0 = tid / PA;
    const int pb = tid % PB, rb0 = tid / PB;
    const bool a_ok = co0 + pa * 8 < d.Cout;
    const uint16_t* const srca = a_ok ? dy + co0 + pa * 8 : zero;
    const unsigned lda = a_ok ? (unsigned)d.ldy : 0u;
    bool b_ok = ci0 + pb * 8 < d.Cin;
    int chb = ci0 + pb * 8, khb = kh, kwb = kw;
    if (tap_fold) {                                        // this piece slot is tap `t` of the group
        const int t = tap * tap_fold + pb;
        khb = t / d.KW;
        kwb = t - khb * d.KW;
        chb = 0;
        b_ok = t < d.KH * d.KW;
    }
    const uint16_t* const srcb = b_ok ? x + chb : zero;
    const unsigned ldb = b_ok ? (unsigned)d.ldx : 0u;
    const stp3_buffer abuf = make_buffer(dy, BUF ? (uint32_t)((size_t)d.M * d.ldy * 2) : 0u);
    const stp3_buffer bbuf = make_buffer(x, BUF ? (uint32_t)((size_t)d.N * d.H * d.W * d.ldx * 2) : 0u);
    const uint32_t cola = (uint32_t)(co0 + pa * 8) * 2u, colb = (uint32_t)chb * 2u;     // BUF: byte offset of the piece in a row
    const uint32_t pitcha = (uint32_t)d.ldy * 2u, pitchb = (uint32_t)d.ldx * 2u;
    const int fold_dh = khb * d.dil_h, fold_dw = kwb * d.dil_w;              // (fold mode) this thread's tap shift

    // ---- pixel role: lane l of EVERY wave owns pixel 64 * step + l of the step being loaded.  (n, ho, wo) is set up with
    // one div / mod pair and advanced by 64 = qa * Ho * Wo + qb * Wo + qc pixels per step with two carries
    const int hw = d.Ho * d.Wo;
    const int qa = kBK / hw, qr = kBK - qa * hw, qb = qr / d.Wo, qc = qr - qb * d.Wo;
    int p_wo, p_ho, p_n;
    {
        const int m = step0 * kBK + lane;
        p_wo = m % d.Wo;
        const int t = m / d.Wo;
        p_ho = t % d.Ho;
        p_n = t / d.Ho;
    }
    const int tap_h = kh * d.dil_h - d.pad_h, tap_w = kw * d.dil_w - d.pad_w;

    u32x4 rawa[NA], rawb[NB];
    auto load_step = [&](int step) {                       // steps must be requested in ascending order, each once
        const int mbase = step * kBK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mbase + ra0 + RA * i;
            if (BUF) rawa[i] = buffer_load16(abuf, (a_ok & (m < d.M)) ? (uint32_t)m * pitcha + cola : kBufOob);
            else rawa[i] = *reinterpret_cast<const u32x4*>(m < d.M ? srca + (size_t)(unsigned)m * lda : zero);
        }
        const bool inside = mbase + lane < d.M;
        if (!tap_fold) {
            const int hi = p_ho * d.stride + tap_h, wi = p_wo * d.stride + tap_w;
            const bool ok = inside && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
            const int pix = ok ? (p_n * d.H + hi) * d.W + wi : -1;          // < 2^31 (checked by the launcher)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int pv = __shfl(pix, rb0 + RB * i);
                if (BUF) rawb[i] = buffer_load16(bbuf, (b_ok & (pv >= 0)) ? (uint32_t)pv * pitchb + colb : kBufOob);
                else rawb[i] = *reinterpret_cast<const u32x4*>(pv >= 0 ? srcb + (size_t)(unsigned)pv * ldb : zero);
            }
        } else {
            const int hs = p_ho * d.stride - d.pad_h, ws = p_wo * d.stride - d.pad_w;
            const int base = (p_n * d.H + hs) * d.W + ws;
            const int pack = inside ? ((hs + 32768) << 16) | (ws + 32768) : 0;   // 0: row -32768, never inside
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int bs = __shfl(base, rb0 + RB * i), pk = __shfl(pack, rb0 + RB * i);
                const int hi = (int)((unsigned)pk >> 16) - 32768 + fold_dh, wi = (pk & 0xffff) - 32768 + fold_dw;
                const bool ok = (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
                const int pv = bs + fold_dh * d.W + fold_dw;
                if (BUF) rawb[i] = buffer_load16(bbuf, (b_ok & ok) ? (uint32_t)pv * pitchb + colb : kBufOob);
                else rawb[i] = *reinterpret_cast<const u32x4*>(ok ? srcb + (size_t)(unsigned)pv * ldb : zero);
            }
        }
        p_wo += qc; p_ho += qb; p_n += qa;                 // this lane's pixel of the next step
        if (p_wo >= d.Wo) { p_wo -= d.Wo; ++p_ho; }
        if (p_ho >= d.Ho) { p_ho -= d.Ho; ++p_n; }
    };
    auto store_step = [&](uint8_t* buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(buf + wg_piece<TCO>(ra0 + RA * i, pa)) = rawa[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(buf + kImgA + wg_piece<TCI>(rb0 + RB * i, pb)) = rawb[i];
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- fragment addresses: lane l reads, of the 16-pixel sub-step ks, the pixel rows 16 ks + 8 (l >> 5) + 4 r +
    // ((l & 15) >> 2), r = 0, 1, and of its 32-channel block the channels 16 ((l >> 4) & 1) + 4 (l & 3) .. + 3; the transpose
    // hands it channel (l & 31) of the rows 8 (l >> 5) .. + 7: an MFMA fragment.  ks and r move the address by whole
    // multiples of 4 rows, which the piece permutation does not see: immediates.
    const int frow = lane & 31, fk = lane >> 5;
    const int fm = 8 * fk + ((lane & 15) >> 2), fc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int offa[TA], offb[TB];
#pragma unroll
    for (int a = 0; a < TA; ++a) {
        const int c = wa * (32 * TA) + a * 32 + fc;
        offa[a] = wg_piece<TCO>(fm, c >> 3) + (c & 7) * 2;
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        const int c = wb * (32 * TB) + b * 32 + fc;
        offb[b] = kImgA + wg_piece<TCI>(fm, c >> 3) + (c & 7) * 2;
    }

    if (step0 < step1) {
        load_step(step0);
        store_step(smem);
    }
    __syncthreads();
    for (int s = step0; s < step1; ++s) {
        const uint8_t* cur = smem + ((s - step0) & 1) * kStage;
        if (s + 1 < step1) load_step(s + 1);
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            FragTr fa[TA], fb[TB];
#pragma unroll
            for (int a = 0; a < TA; ++a) {
                fa[a].h[0] = lds_read_tr16(cur + offa[a] + (ks * 16) * (TCO * 2));
                fa[a].h[1] = lds_read_tr16(cur + offa[a] + (ks * 16 + 4) * (TCO * 2));
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                fb[b].h[0] = lds_read_tr16(cur + offb[b] + (ks * 16) * (TCI * 2));
                fb[b].h[1] = lds_read_tr16(cur + offb[b] + (ks * 16 + 4) * (TCI * 2));
            }
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0);
        }
        if (s + 1 < step1) store_step(smem + ((s + 1 - step0) & 1) * kStage);
        __syncthreads();
    }

    // D[row = co: 8*(r>>2) + 4*(lane>>5) + (r&3)][col = ci: lane & 31]
    const size_t wsize = (size_t)d.Cout * d.KH * d.KW * d.Cin;
    float* out = partial + (size_t)split * wsize;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int ci = ci0 + wb * (32 * TB) + b * 32 + frow;
            // folded: column = (tap, ci) pair number tap * TCI + ci of this tap group, contiguous in dW (Cin == 8)
            const int col = tap_fold ? tap * TCI + ci : tap * d.Cin + ci;
            if (tap_fold ? col >= d.KH * d.KW * d.Cin : ci >= d.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wa * (32 * TA) + a * 32 + 8 * (r >> 2) + 4 * fk + (r & 3);
                if (co < d.Cout) out[(size_t)co * d.KH * d.KW * d.Cin + col] = acc[a][b][r];
            }
        }
}

// dw[i] = sum_k partial[k][i]: 64 columns x 4 split lanes per workgroup (256-byte coalesced row segments, four loads in
// flight per thread), fixed summation order (deterministic)
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(int splits, size_t n, const float* __restrict__ partial,
                                                                  float* __restrict__ dw) {
    __shared__ float red[256];
    const int il = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + il;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int k = kl;
        for (; k + 12 < splits; k += 16) {
            const float a = partial[(size_t)k * n + i], b = partial[(size_t)(k + 4) * n + i];
            const float c = partial[(size_t)(k + 8) * n + i], d = partial[(size_t)(k + 12) * n + i];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; k < splits; k += 4) s0 += partial[(size_t)k * n + i];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (kl == 0 && i < n) dw[i] = (red[il] + red[64 + il]) + (red[128 + il] + red[192 + il]);
}

// The split-K sums of MANY layers in one launch.  Nothing reads the weight gradient of a leaf parameter before the optimizer,
// so the ~100 per-layer reduction launches of a training step (4-7 us each, pure launch floor) are deferred: every layer's
// main kernel leaves its partials in its own slice of an arena, and at the end of the backward pass ONE launch sums them
// all -- job j owns the workgroups [first[j], first[j + 1]), the same four split lanes per column, the same order of
// additions as conv2d_wgrad_reduce_kernel (bit-identical).  The job table travels as the kernel argument (no upload).
constexpr int kWgradBatch = STP3_WGRAD_BATCH_MAX;
struct WgradJobs {
    const float* partial[kWgradBatch];
    float* dw[kWgradBatch];
    unsigned n[kWgradBatch];
    int splits[kWgradBatch];
    int first[kWgradBatch + 1];
    int count;
};
// A workgroup owns 256 columns: thread (column group cg, split lane kl) adds the split slices kl, kl + 4, ... of ITS four
// consecutive columns, 16 bytes per load (the first version read 4 bytes per lane: 1 TB/s on what is a plain stream -- 140 us
// per launch, two launches per step).  Per column the additions are the ones listed above, in the same order.
constexpr int kWgradBatchCols = 256;
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_batch_kernel(WgradJobs jobs) {
    __shared__ float4 red[256];
    // which job: binary search of the workgroup index in first[] (uniform over the workgroup: scalar code)
    int lo = 0, hi = jobs.count;
    const int b = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (jobs.first[mid] <= b) lo = mid; else hi = mid;
    }
    const float* __restrict__ partial = jobs.partial[lo];
    float* __restrict__ dw = jobs.dw[lo];
    const size_t n = jobs.n[lo];
    const int splits = jobs.splits[lo];
    const int cg = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const size_t i = (size_t)(b - jobs.first[lo]) * kWgradBatchCols + 4 * (size_t)cg;
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    if (i + 3 < n && (n & 3) == 0 && ((uintptr_t)partial & 15) == 0) {
        int k = kl;
        for (; k + 12 < splits; k += 16) {
            const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)k * n + i);
            const float4 bb = *reinterpret_cast<const float4*>(partial + (size_t)(k + 4) * n + i);
            const float4 c = *reinterpret_cast<const float4*>(partial + (size_t)(k + 8) * n + i);
            const float4 d = *reinterpret_cast<const float4*>(partial + (size_t)(k + 12) * n + i);
            add(s0, a); add(s1, bb); add(s2, c); add(s3, d);
        }
        for (; k < splits; k += 4) add(s0, *reinterpret_cast<const float4*>(partial + (size_t)k * n + i));
    } else if (i < n) {
        // a ragged end or an odd size: the same sums element by element
        float* s0e = &s0.x; float* s1e = &s1.x; float* s2e = &s2.x; float* s3e = &s3.x;
        for (int e = 0; e < 4 && i + e < n; ++e) {
            int k = kl;
            for (; k + 12 < splits; k += 16) {
                s0e[e] += partial[(size_t)k * n + i + e];       s1e[e] += partial[(size_t)(k + 4) * n + i + e];
                s2e[e] += partial[(size_t)(k + 8) * n + i + e]; s3e[e] += partial[(size_t)(k + 12) * n + i + e];
            }
            for (; k < splits; k += 4) s0e[e] += partial[(size_t)k * n + i + e];
        }
    }
    float4 t;
    t.x = (s0.x + s1.x) + (s2.x + s3.x); t.y = (s0.y + s1.y) + (s2.y + s3.y);
    t.z = (s0.z + s1.z) + (s2.z + s3.z); t.w = (s0.w + s1.w) + (s2.w + s3.w);
    red[threadIdx.x] = t;
    __syncthreads();
    if (kl == 0 && i < n) {
        const float4 a = red[cg], bb = red[64 + cg], c = red[128 + cg], d = red[192 + cg];
        float4 o;
        o.x = (a.x + bb.x) + (c.x + d.x); o.y = (a.y + bb.y) + (c.y + d.y);
        o.z = (a.z + bb.z) + (c.z + d.z); o.w = (a.w + bb.w) + (c.w + d.w);
        if (i + 3 < n && ((uintptr_t)(dw + i) & 15) == 0) {
            *reinterpret_cast<float4*>(dw + i) = o;
        } else {
            const float* oe = &o.x;
            for (int e = 0; e < 4 && i + e < n; ++e) dw[i + e] = oe[e];
        }
    }
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline size_t igemm_lds(int bn, int steps, bool out_f32) {
    const size_t stage = (size_t)(steps > 1 ? 2 : 1) * (kBM + bn) * kBK * 2;   // staging buffers
    // epilogue image [128][bn + 8] in the output type, + the statistics scratch behind a bf16 image
    const size_t tile = out_f32 ? (size_t)kBM * (bn + 8) * 4 : (size_t)kBM * (bn + 8) * 2 + 512 * sizeof(float);
    return stage > tile ? stage : tile;
}

template <int TCO, int TCI>
int wgrad_launch(const ConvDims& d, int tco, int tci, int taps, int splits, int ksteps, int fold, const void* dy,
                 const void* x, void* workspace, hipStream_t s) {
    const size_t lds = (size_t)2 * (TCO + TCI) * kBK * 2;
    // XCD-contiguous workgroup order when the grid is a 32-bit count and the layer has at most a 3x3's worth of taps
    const int xcd = taps <= 9 && (int64_t)tco * tci * taps * splits < (1LL << 31);
    const bool buf = (size_t)d.M * d.ldy * 2 < (1ull << 31) && (size_t)d.N * d.H * d.W * d.ldx * 2 < (1ull << 31);
    if (buf) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wgrad_kernel<TCO, TCI, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        hipLaunchKernelGGL((conv2d_wgrad_kernel<TCO, TCI, true>), dim3(tco * tci, taps, splits), dim3(256), lds, s, d, tci,
                           ksteps, fold, xcd, (const uint16_t*)dy, (const uint16_t*)x, (float*)workspace);
        return STP3_OK;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wgrad_kernel<TCO, TCI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL((conv2d_wgrad_kernel<TCO, TCI>), dim3(tco * tci, taps, splits), dim3(256), lds, s, d, tci, ksteps,
                       fold, xcd, (const uint16_t*)dy, (const uint16_t*)x, (float*)workspace);
    return STP3_OK;
}

}  // namespace

extern "C" {

int stp3_conv2d_fwd_workspace(const stp3_conv_dims* p, size_t* bytes) {
    if (!p || !bytes || p->N <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->Cout <= 0) return STP3_EINVAL;
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    const size_t parts = (size_t)((M + kBM - 1) / kBM);
    if (parts > 65536) return STP3_EUNSUP;
    *bytes = (parts + (parts + 255) / 256) * 2 * p->Cout * sizeof(float);
    return STP3_OK;
}

}  // extern "C"

namespace {

template <int BN, int MODE, int ACT>
int igemm_launch_one(const ConvDims& d, dim3 grid, size_t lds, int tiles_co, const void* x, const void* w, const float* bias,
                     void* y, float* partial, const EpiArgs& ep, hipStream_t s) {
    // input and weights each below 2 GiB: buffer-addressed staging (see the kernel); the stress geometry's 4.6 G-element
    // tensors keep the 64-bit addresses
    const bool buf = (size_t)d.N * d.H * d.W * d.ldx * 2 < (1ull << 31) && (size_t)d.Cout * d.Ktot * 2 < (1ull << 31);
    if (buf) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_igemm_kernel<BN, MODE, ACT, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        hipLaunchKernelGGL((conv2d_igemm_kernel<BN, MODE, ACT, true>), grid, dim3(256), lds, s, d, tiles_co, (const uint16_t*)x,
                           (const uint16_t*)w, bias, y, partial, ep);
        return STP3_OK;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_igemm_kernel<BN, MODE, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL((conv2d_igemm_kernel<BN, MODE, ACT>), grid, dim3(256), lds, s, d, tiles_co, (const uint16_t*)x,
                       (const uint16_t*)w, bias, y, partial, ep);
    return STP3_OK;
}

template <int BN, int MODE>
int igemm_launch_act(int act, const ConvDims& d, dim3 grid, size_t lds, int tiles_co, const void* x, const void* w,
                     const float* bias, void* y, float* partial, const EpiArgs& ep, hipStream_t s) {
    if (MODE == kModePlain || MODE == kModeStats || act == STP3_ACT_NONE)
        return igemm_launch_one<BN, MODE, STP3_ACT_NONE>(d, grid, lds, tiles_co, x, w, bias, y, partial, ep, s);
    if (act == STP3_ACT_RELU) return igemm_launch_one<BN, MODE, STP3_ACT_RELU>(d, grid, lds, tiles_co, x, w, bias, y, partial, ep, s);
    return igemm_launch_one<BN, MODE, STP3_ACT_SWISH>(d, grid, lds, tiles_co, x, w, bias, y, partial, ep, s);
}

template <int PP, int KMAX, int MODE>
int pointwise_rows_launch_one(const ConvDims& d, int ksteps, int tiles_co, unsigned nwg, size_t lds, const void* x, const void* w,
                              void* y, float* partial, const EpiArgs& ep, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pointwise_rows_kernel<PP, KMAX, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL((pointwise_rows_kernel<PP, KMAX, MODE>), dim3(nwg), dim3(256), lds, s, d, ksteps, tiles_co,
                       (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, partial, ep);
    return STP3_OK;
}

template <int PP, int KMAX>
int pointwise_rows_launch_mode(int mode, const ConvDims& d, int ksteps, int tiles_co, unsigned nwg, size_t lds, const void* x,
                               const void* w, void* y, float* partial, const EpiArgs& ep, hipStream_t s) {
    switch (mode) {
        case kModePlain: return pointwise_rows_launch_one<PP, KMAX, kModePlain>(d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        case kModeStats: return pointwise_rows_launch_one<PP, KMAX, kModeStats>(d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        case kModeBnAct: return pointwise_rows_launch_one<PP, KMAX, kModeBnAct>(d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        case kModeBwdReduce:
            return pointwise_rows_launch_one<PP, KMAX, kModeBwdReduce>(d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        default: return pointwise_rows_launch_one<PP, KMAX, kModeBwdApply>(d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
    }
}

// one wave per (pixel group, 32-channel block); the grid is ONE resident round of this instantiation's workgroups (4 per CU for
// the thin layers' forward modes, 1-3 for the register-heavier ones), never more pixel groups than 128-pixel blocks (the
// partial-sum rows) or 32-pixel tiles
template <int KMAX, int MODE>
int pointwise_direct_launch_one(const ConvDims& d, int ksteps, unsigned gx, int* parts, const void* x, const void* w, void* y,
                                float* partial, const EpiArgs& ep, hipStream_t s) {
    static const int resident = [] {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pointwise_direct_kernel<KMAX, MODE>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        return per_cu * 256;
    }();
    const int cblocks = (d.Cout + 31) / 32;
    const int ntiles = (d.M + 31) / 32;
    int ngroups = resident * 4 / cblocks;
    if (ngroups > (int)gx) ngroups = (int)gx;
    if (ngroups > ntiles) ngroups = ntiles;
    if (ngroups < 1) ngroups = 1;
    *parts = ngroups;
    const unsigned nwg = (unsigned)(((int64_t)ngroups * cblocks + 3) / 4);
    hipLaunchKernelGGL((pointwise_direct_kernel<KMAX, MODE>), dim3(nwg), dim3(256), 0, s, d, ksteps, cblocks, ngroups,
                       (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, partial, ep);
    return STP3_OK;
}

template <int KMAX>
int pointwise_direct_launch_mode(int mode, const ConvDims& d, int ksteps, unsigned gx, int* parts, const void* x, const void* w,
                                 void* y, float* partial, const EpiArgs& ep, hipStream_t s) {
    switch (mode) {
        case kModePlain: return pointwise_direct_launch_one<KMAX, kModePlain>(d, ksteps, gx, parts, x, w, y, partial, ep, s);
        case kModeStats: return pointwise_direct_launch_one<KMAX, kModeStats>(d, ksteps, gx, parts, x, w, y, partial, ep, s);
        case kModeBnAct: return pointwise_direct_launch_one<KMAX, kModeBnAct>(d, ksteps, gx, parts, x, w, y, partial, ep, s);
        case kModeBwdReduce: return pointwise_direct_launch_one<KMAX, kModeBwdReduce>(d, ksteps, gx, parts, x, w, y, partial, ep, s);
        default: return pointwise_direct_launch_one<KMAX, kModeBwdApply>(d, ksteps, gx, parts, x, w, y, partial, ep, s);
    }
}

constexpr int kPointwiseMaxCin = 128;

// the layers the streaming kernels take (see pointwise_rows_kernel): 1x1 / stride 1 / no padding, contraction <= 128, at
// least 64 output channels in whole 16-byte pieces, bf16 output, no bias
bool pointwise_applies(const stp3_conv_dims* p, const void* y) {
    return p->KH == 1 && p->KW == 1 && p->stride == 1 && p->pad_h == 0 && p->pad_w == 0 && p->H == p->Ho && p->W == p->Wo &&
           p->Cin <= kPointwiseMaxCin && p->Cout >= 64 && !p->has_bias && p->out_dtype == STP3_DTYPE_BF16 &&
           p->Cout % 8 == 0 && p->ldy % 8 == 0 && !((uintptr_t)y & 15);
}

int pointwise_run(const ConvDims& d, const void* x, const void* w, void* y, float* sums, float* partial, unsigned gx,
                  hipStream_t s, int mode, const EpiArgs& ep) {
    const int ksteps = (d.Cin + 15) / 16;
    int rc, parts;
    // the kernel of whole pixel rows / whole lines: the 144- and 192-channel layers of the trunk (one channel tile) and every
    // layer of k * 64 channels (tiles of 128 or 64 channels = whole 128-byte lines; e.g. the 128 -> 512 data gradient of the
    // temporal model's ASPP projection, 491 MB: 314 us in 32-byte pieces, 246 us on the tiled kernel, 143 us here; and the
    // in-cache 64 -> 128 @200x200x12: 46 / 36 / 31 us)
    const bool rows_small = (d.Cout == 144 || d.Cout == 192) && ksteps <= 2;
    const bool rows_lines = d.Cout % 64 == 0;
    if ((rows_small || rows_lines) && d.ldy % 8 == 0 && (mode < kModeBwdReduce || ep.ldz % 8 == 0)) {
        const int pp = rows_small ? d.Cout / 8 : (d.Cout % 128 == 0 ? 16 : 8), nt = (pp + 3) / 4;
        const int tiles_co = d.Cout / (pp * 8);
        if (ep.dx && !(rows_small && mode == kModeBwdApply && d.Cin <= 32)) return STP3_EUNSUP;
        const size_t lds = (size_t)nt * 32 * (ksteps * 16 + 8) * 2 + (size_t)4 * 32 * (nt * 32 + 8) * 2 +
                           (ep.dx ? (size_t)32 * (nt * 32 + 8) * 2 : 0);
        // persistent: as many workgroups per CU as their LDS (weight image + four wave tiles) allows, at most 3
        unsigned per_cu = (unsigned)((160 * 1024) / lds);
        if (per_cu > 3) per_cu = 3;
        unsigned groups = (256 * per_cu + tiles_co - 1) / tiles_co;
        if (groups > gx) groups = gx;
        parts = (int)groups;
        const unsigned nwg = groups * (unsigned)tiles_co;
        if (pp == 18) rc = pointwise_rows_launch_mode<18, 2>(mode, d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        else if (pp == 24) rc = pointwise_rows_launch_mode<24, 2>(mode, d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        else if (pp == 8) rc = pointwise_rows_launch_mode<8, 8>(mode, d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        else if (ksteps <= 4) rc = pointwise_rows_launch_mode<16, 4>(mode, d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
        else rc = pointwise_rows_launch_mode<16, 8>(mode, d, ksteps, tiles_co, nwg, lds, x, w, y, partial, ep, s);
    } else {
        if (ep.dx) return STP3_EUNSUP;
        if (ksteps <= 2) rc = pointwise_direct_launch_mode<2>(mode, d, ksteps, gx, &parts, x, w, y, partial, ep, s);
        else if (ksteps <= 4) rc = pointwise_direct_launch_mode<4>(mode, d, ksteps, gx, &parts, x, w, y, partial, ep, s);
        else rc = pointwise_direct_launch_mode<8>(mode, d, ksteps, gx, &parts, x, w, y, partial, ep, s);
    }
    if (rc) return rc;
    if (sums) launch_colsum(s, parts, 2 * d.Cout, partial, sums);
    return status();
}

// the forward kernel in one of its epilogue modes (see kMode*): shared argument checks, tile choice and launch
int igemm_run(const stp3_conv_dims* p, const void* x, const void* w, const float* bias, void* y, float* sums, void* workspace,
              size_t workspace_bytes, void* stream, int mode, int act, EpiArgs ep) {
    if (!p || !x || !w || (mode != kModeStats && mode != kModeBwdReduce && !y)) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->KH <= 0 ||
        p->KW <= 0 || p->stride <= 0 || p->dil_h <= 0 || p->dil_w <= 0 || p->pad_h < 0 || p->pad_w < 0)
        return STP3_EINVAL;
    if (p->has_bias && !bias) return STP3_EINVAL;
    if (p->Cin % 8 || p->ldx % 8 || p->ldx < p->Cin || p->ldy < p->Cout) return STP3_EUNSUP;   // 16-byte k pieces
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return STP3_EUNSUP;
    if (p->out_dtype != STP3_DTYPE_BF16 && p->out_dtype != STP3_DTYPE_F32) return STP3_EUNSUP;
    if (y && ((uintptr_t)y & (p->out_dtype == STP3_DTYPE_F32 ? 3 : 1))) return STP3_EUNSUP;
    if ((sums || mode != kModePlain) && p->out_dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (mode != kModePlain) {
        // the fused-BatchNorm modes: whole 16-byte channel pieces on every side, no convolution bias
        if (p->has_bias || p->Cout % 8 || p->ldy % 8 || (y && ((uintptr_t)y & 15))) return STP3_EUNSUP;
        if (mode >= kModeBnAct && !ep.coef) return STP3_EINVAL;
        if (mode >= kModeBwdReduce && (!ep.dz || ep.ldz % 8 || ep.ldz < p->Cout || ((uintptr_t)ep.dz & 15))) return STP3_EUNSUP;
        if (mode == kModeBwdApply && !ep.gsums) return STP3_EINVAL;
        if ((mode == kModeStats || mode == kModeBwdReduce) && !sums) return STP3_EINVAL;
    }
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    // 32-bit PIXEL indices (input and output); element offsets are 64-bit pointer arithmetic since the round-3 staging
    if (M >= (1LL << 31) || (int64_t)p->N * p->H * p->W >= (1LL << 31)) return STP3_EUNSUP;
    ConvDims d;
    d.N = p->N; d.H = p->H; d.W = p->W; d.Cin = p->Cin; d.Ho = p->Ho; d.Wo = p->Wo; d.Cout = p->Cout;
    d.KH = p->KH; d.KW = p->KW; d.stride = p->stride; d.pad_h = p->pad_h; d.pad_w = p->pad_w;
    d.dil_h = p->dil_h; d.dil_w = p->dil_w; d.ldx = p->ldx; d.ldy = p->ldy;
    d.out_f32 = p->out_dtype == STP3_DTYPE_F32; d.has_bias = p->has_bias;
    d.M = (int)M; d.kchunks = (p->Cin + 31) / 32; d.Ktot = p->KH * p->KW * p->Cin;
    const unsigned gx = (unsigned)((M + kBM - 1) / kBM);
    hipStream_t s = (hipStream_t)stream;
    float* partial = nullptr;
    if (sums) {
        if (gx > 65536) return STP3_EUNSUP;
        if (!workspace || workspace_bytes < ((size_t)gx + (gx + 255) / 256) * 2 * p->Cout * sizeof(float)) return STP3_ENOSPACE;
        partial = (float*)workspace;
    }
    if (y && ((uintptr_t)y & 15)) return STP3_EUNSUP;
    if (ep.dx) {
        // BWD_APPLY + data gradient: the whole-row streaming kernel only (stp3_conv2d_bn_bwd_apply_dx)
        if (!pointwise_applies(p, y) || ep.lddx % 4 || ep.lddx < p->Cin || ((uintptr_t)ep.dx & 7) ||
            (ep.add && (ep.ldadd % 4 || ep.ldadd < p->Cin || ((uintptr_t)ep.add & 7))))
            return STP3_EUNSUP;
        ep.act = act;
        return pointwise_run(d, x, w, y, sums, partial, gx, s, mode, ep);
    }
    if (ep.add) {
        // the skip-gradient addend: plain mode, bf16 output, whole 16-byte channel pieces on both sides
        if (mode != kModePlain || d.out_f32 || p->Cout % 8 || p->ldy % 8 || ep.ldadd % 8 || ep.ldadd < p->Cout ||
            ((uintptr_t)ep.add & 15) || ((uintptr_t)y & 15))
            return STP3_EUNSUP;
    } else if (pointwise_applies(p, y)) {
        ep.act = act;
        return pointwise_run(d, x, w, y, sums, partial, gx, s, mode, ep);
    }
    // 128 x 128 tiles for the contraction-heavy layers; 128 x 64 tiles when there are few output channels or so little K
    // (<= 2 steps: the pointwise layers of the trunk, bound by activation traffic) that what counts is many light
    // workgroups per CU; one staging buffer suffices for a single K step
    const int steps = (d.Ktot + kBK - 1) / kBK;
    // ... and the narrow tile too when the wide tiling would leave the chip under-filled (fewer than two workgroups per CU: the
    // 25 x 25 and 50 x 50 stages of the decoder, the 14 x 30 maps of the trunk -- 118 / 235 / 237 workgroups of the wide tile):
    // twice the workgroups hide each other's latency (round 6: -0.15 ms per step, profiles/r06y_igemm_narrow_tile.txt)
    const bool wide = p->Cout > 64 && steps > 2 && (int64_t)gx * ((p->Cout + 127) / 128) >= 512;
    const int bn = wide ? 128 : 64;
    size_t lds = igemm_lds(bn, steps, d.out_f32 != 0);
    if (mode == kModeBwdReduce) {                             // output tile + gradient tile + the reduction scratch
        const size_t need = (size_t)2 * kBM * (bn + 8) * 2 + 512 * sizeof(float);
        if (need > lds) lds = need;
    }
    const int tiles_co = (p->Cout + bn - 1) / bn;
    if ((int64_t)gx * tiles_co >= (1LL << 31)) return STP3_EUNSUP;
    const dim3 grid(gx * (unsigned)tiles_co);                // 1-D: the kernel derives (pixel tile, channel tile) itself
    int rc;
#define STP3_IGEMM_MODE(MODE)                                                                                                \
    rc = wide ? igemm_launch_act<128, MODE>(act, d, grid, lds, tiles_co, x, w, bias, y, partial, ep, s)                       \
              : igemm_launch_act<64, MODE>(act, d, grid, lds, tiles_co, x, w, bias, y, partial, ep, s)
    switch (mode) {
        case kModePlain: STP3_IGEMM_MODE(kModePlain); break;
        case kModeStats: STP3_IGEMM_MODE(kModeStats); break;
        case kModeBnAct: STP3_IGEMM_MODE(kModeBnAct); break;
        case kModeBwdReduce: STP3_IGEMM_MODE(kModeBwdReduce); break;
        default: STP3_IGEMM_MODE(kModeBwdApply); break;
    }
#undef STP3_IGEMM_MODE
    if (rc) return rc;
    if (sums) launch_colsum(s, (int)gx, 2 * p->Cout, partial, sums);
    return status();
}

}  // namespace

extern "C" {

int stp3_conv2d_fwd(const stp3_conv_dims* p, const void* x, const void* w, const float* bias, void* y, float* sums,
                    void* workspace, size_t workspace_bytes, void* stream) {
    return igemm_run(p, x, w, bias, y, sums, workspace, workspace_bytes, stream, kModePlain, STP3_ACT_NONE, EpiArgs());
}

int stp3_conv2d_fwd_add(const stp3_conv_dims* p, const void* x, const void* w, const float* bias, const void* add, int32_t ldadd,
                        void* y, void* stream) {
    if (!add) return STP3_EINVAL;
    EpiArgs ep = EpiArgs();
    ep.add = (const uint16_t*)add;
    ep.ldadd = ldadd;
    return igemm_run(p, x, w, bias, y, nullptr, nullptr, 0, stream, kModePlain, STP3_ACT_NONE, ep);
}

// ---- convolution -> BatchNorm -> activation WITHOUT the convolution output in memory (see kMode*) -------------------
int stp3_conv2d_fwd_stats(const stp3_conv_dims* p, const void* x, const void* w, float* sums, void* workspace,
                          size_t workspace_bytes, void* stream) {
    return igemm_run(p, x, w, nullptr, nullptr, sums, workspace, workspace_bytes, stream, kModeStats, STP3_ACT_NONE, EpiArgs());
}

int stp3_conv2d_fwd_bnact(const stp3_conv_dims* p, const void* x, const void* w, const float* coef, int32_t act, void* y,
                          void* stream) {
    EpiArgs ep = EpiArgs();
    ep.coef = coef;
    return igemm_run(p, x, w, nullptr, y, nullptr, nullptr, 0, stream, kModeBnAct, act, ep);
}

int stp3_conv2d_bn_bwd_reduce(const stp3_conv_dims* p, const void* x, const void* w, const void* dz, int32_t ldz,
                              const float* coef, int32_t act, float* sums, void* workspace, size_t workspace_bytes,
                              void* stream) {
    EpiArgs ep = EpiArgs();
    ep.coef = coef; ep.dz = (const uint16_t*)dz; ep.ldz = ldz;
    return igemm_run(p, x, w, nullptr, nullptr, sums, workspace, workspace_bytes, stream, kModeBwdReduce, act, ep);
}

int stp3_conv2d_bn_bwd_apply(const stp3_conv_dims* p, const void* x, const void* w, const void* dz, int32_t ldz,
                             const float* coef, int32_t act, const float* gsums, double count, void* dy, void* stream) {
    if (!(count >= 1.0)) return STP3_EINVAL;
    EpiArgs ep = EpiArgs();
    ep.coef = coef; ep.dz = (const uint16_t*)dz; ep.ldz = ldz; ep.gsums = gsums; ep.inv_count = (float)(1.0 / count);
    return igemm_run(p, x, w, nullptr, dy, nullptr, nullptr, 0, stream, kModeBwdApply, act, ep);
}

int stp3_conv2d_bn_bwd_apply_dx(const stp3_conv_dims* p, const void* x, const void* w, const void* dz, int32_t ldz,
                                const float* coef, int32_t act, const float* gsums, double count, void* dy, void* dx,
                                int32_t lddx, const void* add, int32_t ldadd, void* stream) {
    if (!(count >= 1.0) || !dx) return STP3_EINVAL;
    EpiArgs ep = EpiArgs();
    ep.coef = coef; ep.dz = (const uint16_t*)dz; ep.ldz = ldz; ep.gsums = gsums; ep.inv_count = (float)(1.0 / count);
    ep.dx = (uint16_t*)dx; ep.lddx = lddx; ep.add = (const uint16_t*)add; ep.ldadd = ldadd;
    return igemm_run(p, x, w, nullptr, dy, nullptr, nullptr, 0, stream, kModeBwdApply, act, ep);
}


// *fold: taps per workgroup when the taps are folded into the ci tile (Cin == 8 with more than one tap), else 0;
// *grid_y: tap groups (folded) or taps
static int wgrad_plan(const stp3_conv_dims* p, int* tco_sz, int* tci_sz, int* tiles_co, int* tiles_ci, int* splits, int* ksteps,
                      int* fold, int* grid_y) {
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    const int taps = p->KH * p->KW;
    *tco_sz = p->Cout > 64 ? 128 : 64;
    *tci_sz = p->Cin > 64 ? 128 : 64;
    *fold = 0;
    *grid_y = taps;
    if (p->Cin == 8 && taps > 1) {
        *tci_sz = taps > 8 ? 128 : 64;
        *fold = *tci_sz / 8;
        *grid_y = (taps + *fold - 1) / *fold;
    }
    *tiles_co = (p->Cout + *tco_sz - 1) / *tco_sz;
    *tiles_ci = *fold ? 1 : (p->Cin + *tci_sz - 1) / *tci_sz;
    const int64_t total_steps = (M + kBK - 1) / kBK;         // 64 pixels per step
    const int64_t base = (int64_t)(*tiles_co) * (*tiles_ci) * (*grid_y);
    // AT MOST one resident round of workgroups: conv2d_wgrad_kernel<128, 128> keeps 2 per CU (224 registers, 64 KB of LDS),
    // <128, 64> and <64, 128> 3 (144, 48 KB), <64, 64> 5 (96, 32 KB) -- every workgroup walks the same number of K-steps, so
    // the ~1024 (rounded up) of rounds 2-3 were 2.004 rounds of 512 places for a 3x3 128 -> 128 layer: a third round for 2
    // workgroups (tests/test_kernel_resources_cpu.py pins the register budgets)
    const int per_cu = (*tco_sz == 128 && *tci_sz == 128) ? 2 : (*tco_sz == 128 || *tci_sz == 128) ? 3 : 5;
    int64_t want = (256 * per_cu) / base;
    const int64_t max_splits = (total_steps + 7) / 8;        // at least 8 k-steps per workgroup
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    *ksteps = (int)((total_steps + want - 1) / want);
    *splits = (int)((total_steps + *ksteps - 1) / *ksteps);
    return STP3_OK;
}

int stp3_conv2d_wgrad_workspace(const stp3_conv_dims* p, size_t* bytes) {
    if (!p || !bytes || p->N <= 0 || p->Cout <= 0 || p->Cin <= 0 || p->KH <= 0 || p->KW <= 0 || p->Ho <= 0 || p->Wo <= 0)
        return STP3_EINVAL;
    int a, b, tco, tci, splits, ksteps, fold, gy;
    wgrad_plan(p, &a, &b, &tco, &tci, &splits, &ksteps, &fold, &gy);
    *bytes = (size_t)splits * p->Cout * p->KH * p->KW * p->Cin * sizeof(float);
    return STP3_OK;
}

// the main kernel of the weight gradient: partial[split][Cout][KH][KW][Cin] into `workspace`; *splits_out = the splits
static int wgrad_partials(const stp3_conv_dims* p, const void* dy, const void* x, void* workspace, size_t workspace_bytes,
                          int* splits_out, hipStream_t s) {
    if (!p || !dy || !x || !workspace) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->KH <= 0 ||
        p->KW <= 0 || p->stride <= 0 || p->dil_h <= 0 || p->dil_w <= 0 || p->pad_h < 0 || p->pad_w < 0)
        return STP3_EINVAL;
    // 16-byte channel blocks of both operands
    if (p->Cin % 8 || p->Cout % 8 || p->ldx % 8 || p->ldy % 8 || p->ldx < p->Cin || p->ldy < p->Cout) return STP3_EUNSUP;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15)) return STP3_EUNSUP;
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    // 32-bit pixel indices (the pixel lanes of the last step run up to 63 pixels past the end: two images of slack), and
    // 16-bit packed (row, column) of the tap-folded path
    if (M >= (1LL << 31) - 64 || ((int64_t)p->N + 2) * p->H * p->W >= (1LL << 31)) return STP3_EUNSUP;
    if ((int64_t)p->Ho * p->stride >= 32768 || (int64_t)p->Wo * p->stride >= 32768 || p->pad_h >= 32768 || p->pad_w >= 32768)
        return STP3_EUNSUP;
    if (p->KH * p->KW > 65535) return STP3_EUNSUP;
    int tco_sz, tci_sz, tco, tci, splits, ksteps, fold, taps;
    wgrad_plan(p, &tco_sz, &tci_sz, &tco, &tci, &splits, &ksteps, &fold, &taps);
    const size_t wsize = (size_t)p->Cout * p->KH * p->KW * p->Cin;
    if (workspace_bytes < (size_t)splits * wsize * sizeof(float)) return STP3_ENOSPACE;
    ConvDims d;
    d.N = p->N; d.H = p->H; d.W = p->W; d.Cin = p->Cin; d.Ho = p->Ho; d.Wo = p->Wo; d.Cout = p->Cout;
    d.KH = p->KH; d.KW = p->KW; d.stride = p->stride; d.pad_h = p->pad_h; d.pad_w = p->pad_w;
    d.dil_h = p->dil_h; d.dil_w = p->dil_w; d.ldx = p->ldx; d.ldy = p->ldy;
    d.out_f32 = 1; d.has_bias = 0; d.M = (int)M; d.kchunks = 0; d.Ktot = p->KH * p->KW * p->Cin;
    int rc;
    if (tco_sz == 128 && tci_sz == 128) rc = wgrad_launch<128, 128>(d, tco, tci, taps, splits, ksteps, fold, dy, x, workspace, s);
    else if (tco_sz == 128) rc = wgrad_launch<128, 64>(d, tco, tci, taps, splits, ksteps, fold, dy, x, workspace, s);
    else if (tci_sz == 128) rc = wgrad_launch<64, 128>(d, tco, tci, taps, splits, ksteps, fold, dy, x, workspace, s);
    else rc = wgrad_launch<64, 64>(d, tco, tci, taps, splits, ksteps, fold, dy, x, workspace, s);
    if (rc) return rc;
    *splits_out = splits;
    return status();
}

int stp3_conv2d_wgrad(const stp3_conv_dims* p, const void* dy, const void* x, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream) {
    if (!dw) return STP3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int splits = 0;
    int rc = wgrad_partials(p, dy, x, workspace, workspace_bytes, &splits, s);
    if (rc) return rc;
    const size_t wsize = (size_t)p->Cout * p->KH * p->KW * p->Cin;
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((wsize + 63) / 64)), dim3(256), 0, s, splits, wsize,
                       (const float*)workspace, dw);
    return status();
}

int stp3_conv2d_wgrad_partials(const stp3_conv_dims* p, const void* dy, const void* x, void* partials, size_t partials_bytes,
                               int32_t* splits, void* stream) {
    if (!splits) return STP3_EINVAL;
    int n = 0;
    int rc = wgrad_partials(p, dy, x, partials, partials_bytes, &n, (hipStream_t)stream);
    if (rc) return rc;
    *splits = n;
    return STP3_OK;
}

int stp3_conv2d_wgrad_reduce_batch(int32_t n, const stp3_wgrad_job* jobs, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return STP3_EINVAL;
    for (int j = 0; j < n; ++j)
        if (!jobs[j].partials || !jobs[j].dw || jobs[j].splits <= 0 || jobs[j].numel <= 0 || jobs[j].numel >= (1LL << 32))
            return STP3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    for (int j0 = 0; j0 < n; j0 += kWgradBatch) {
        WgradJobs t;
        t.count = n - j0 < kWgradBatch ? n - j0 : kWgradBatch;
        int64_t blocks = 0;
        for (int j = 0; j < t.count; ++j) {
            const stp3_wgrad_job& q = jobs[j0 + j];
            t.partial[j] = (const float*)q.partials;
            t.dw[j] = q.dw;
            t.n[j] = (unsigned)q.numel;
            t.splits[j] = q.splits;
            t.first[j] = (int)blocks;
            blocks += (q.numel + kWgradBatchCols - 1) / kWgradBatchCols;
            if (blocks >= (1LL << 31)) return STP3_EUNSUP;
        }
        for (int j = t.count; j <= kWgradBatch; ++j) t.first[j] = (int)blocks;
        for (int j = t.count; j < kWgradBatch; ++j) { t.partial[j] = nullptr; t.dw[j] = nullptr; t.n[j] = 0; t.splits[j] = 0; }
        hipLaunchKernelGGL(conv2d_wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t);
    }
    return status();
}

}  // extern "C"
