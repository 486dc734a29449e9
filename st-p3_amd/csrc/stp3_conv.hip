// stp3_conv.hip -- bf16 MFMA implicit-GEMM 2-D convolution (NHWC) for gfx950.
//
// Replaces the dense nn.Conv2d / frame-folded nn.Conv3d contractions of the reference's hot path:
//   stp3/layers/convolutions.py:183-280 (UpsamplingConcat / UpsamplingAdd / ASPP / DeepLabHead 1x1, 3x3 and
//   dilated 3x3), stp3/layers/temporal.py:252-273, 315-325 (CausalConv3d (2,3,3)/(1,3,3) and 1x1x1, run
//   frame-folded as 2-D convolutions), stp3/models/decoder.py:22-140 (7x7/2 stem, ResNet-18 3x3, heads) and the
//   1x1 expand / project convolutions of the EfficientNet MBConv blocks driven by stp3/models/encoder.py:57-97.
//
// GEMM view:  Y[m][co] = sum_{tap, ci} X[pixel(m) + tap][ci] * W[co][tap][ci],  m = (n, ho, wo).
// Both operands are K-contiguous in memory (NHWC activations, [Cout][KH][KW][Cin] weights), which is exactly
// the v_mfma_f32_16x16x32_bf16 fragment shape: lane l supplies 8 consecutive k of row (l & 15), k-chunk
// (l >> 4) -- one 16-byte load per lane per fragment, no LDS shuffle.  The weights take the A (row) slot and
// the pixels the B (column) slot, so an accumulator lane ends up with 4 consecutive output CHANNELS of one
// pixel: the epilogue stores 8-byte bf16x4 pieces straight into the NHWC output.
//
// Tile: workgroup = 4 waves = 128 pixels x 64 channels; wave = 32 pixels x 64 channels = 2 x 4 MFMA tiles
// (8 accumulators, 32 VGPRs).  Fragments of step s+1 are loaded while step s is multiplied (register double
// buffer).  A k-step is 32 input channels of one tap.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "stp3_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvDims {
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int ldx, ldy;
    int out_f32, has_bias;
    int M;          // N * Ho * Wo
    int kchunks;    // ceil(Cin / 32)
};

union Frag {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ uint32_t f2bf(float a) {   // round to nearest even
    uint32_t u = __float_as_uint(a);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

constexpr int kTilePix = 128;   // pixels per workgroup
constexpr int kTileCo = 64;     // output channels per workgroup
constexpr int PT = 2;           // 16-pixel MFMA tiles per wave
constexpr int CT = 4;           // 16-channel MFMA tiles per wave

__global__ __launch_bounds__(256) void conv2d_fwd_kernel(ConvDims d, const uint16_t* __restrict__ x,
                                                         const uint16_t* __restrict__ w,
                                                         const float* __restrict__ bias, void* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15;       // row of the A fragment (output channel) / column of B (pixel)
    const int kq = lane >> 4;       // k-chunk: 8 input channels
    const int m_wave = blockIdx.x * kTilePix + wave * (PT * 16);
    const int co0 = blockIdx.y * kTileCo;

    // ---- per-lane pixel coordinates of the B fragments ------------------------------------------------
    int pn[PT], ph[PT], pw[PT];
    bool pvalid[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int m = m_wave + j * 16 + li;
        pvalid[j] = m < d.M;
        const int mm = pvalid[j] ? m : 0;
        const int wo = mm % d.Wo;
        const int t = mm / d.Wo;
        const int ho = t % d.Ho;
        pn[j] = t / d.Ho;
        ph[j] = ho * d.stride - d.pad_h;
        pw[j] = wo * d.stride - d.pad_w;
    }
    // ---- per-lane weight rows of the A fragments ------------------------------------------------------
    const size_t wrow = (size_t)d.KH * d.KW * d.Cin;           // elements per output channel
    bool cvalid[CT];
    const uint16_t* wp[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        const int co = co0 + i * 16 + li;
        cvalid[i] = co < d.Cout;
        wp[i] = w + (size_t)(cvalid[i] ? co : 0) * wrow + kq * 8;
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int steps = d.KH * d.KW * d.kchunks;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    // fragments of one k-step: step -> (tap = step / kchunks, c0 = (step % kchunks) * 32)
    auto load_step = [&](int step, Frag (&a)[CT], Frag (&b)[PT]) {
        const int tap = step / d.kchunks;
        const int c0 = (step - tap * d.kchunks) * 32 + kq * 8;
        const bool kvalid = c0 < d.Cin;                        // Cin % 8 == 0: a chunk is all-in or all-out
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            a[i].u = (cvalid[i] && kvalid)
                         ? *reinterpret_cast<const uint4*>(wp[i] + (size_t)tap * d.Cin + (c0 - kq * 8))
                         : zero4;
        }
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int hi = ph[j] + kh * d.dil_h;
            const int wi = pw[j] + kw * d.dil_w;
            const bool ok = pvalid[j] && kvalid && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
            b[j].u = ok ? *reinterpret_cast<const uint4*>(x + ((size_t)(pn[j] * d.H + hi) * d.W + wi) * d.ldx + c0)
                        : zero4;
        }
    };

    Frag a0[CT], b0[PT], a1[CT], b1[PT];
    load_step(0, a0, b0);
    for (int step = 0; step < steps; step += 2) {
        if (step + 1 < steps) load_step(step + 1, a1, b1);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i].v, b0[j].v, acc[i][j], 0, 0, 0);
        if (step + 1 < steps) {
            if (step + 2 < steps) load_step(step + 2, a0, b0);
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i].v, b1[j].v, acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: D[row = channel (lane>>4)*4 + r][col = pixel lane&15] -------------------------------
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int m = m_wave + j * 16 + li;
        if (m >= d.M) continue;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int co = co0 + i * 16 + kq * 4;
            if (co >= d.Cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (d.has_bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < d.Cout) v[r] += bias[co + r];
            }
            const bool full = co + 3 < d.Cout;
            if (d.out_f32) {
                float* yp = (float*)y + (size_t)m * d.ldy + co;
                if (full && ((d.ldy & 3) == 0)) {
                    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < d.Cout) yp[r] = v[r];
                }
            } else {
                uint16_t* yp = (uint16_t*)y + (size_t)m * d.ldy + co;
                if (full && ((d.ldy & 3) == 0)) {
                    *reinterpret_cast<uint2*>(yp) = make_uint2(f2bf(v[0]) | (f2bf(v[1]) << 16), f2bf(v[2]) | (f2bf(v[3]) << 16));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < d.Cout) yp[r] = (uint16_t)f2bf(v[r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[co][tap][ci] = sum_m dY[m][co] * X[pixel(m) + tap][ci]
// ------------------------------------------------------------------------------------------------
// The contraction runs over pixels, the SLOW dimension of both NHWC operands, while an MFMA fragment wants 8
// consecutive k per lane.  Each lane therefore loads 8 consecutive pixels x 4 channels (8 x 8 bytes) per operand
// and transposes the 8x4 block in registers (v_perm_b32): lane (i, kq) ends up with, for each of its 4 channels
// c = 8*i + 4*half + a, the 8 pixels kq*8 .. kq*8+7 -- fragment `a` of an MFMA whose row i stands for channel
// 8*i + 4*half + a.  One k-step (32 pixels) feeds a 128 x 128 (co x ci) tile; the 4 waves of a workgroup own
// the 4 (co half, ci half) quadrants: 4 x 4 MFMAs and 64 accumulator registers each.
// Pixels are split over gridDim.z workgroups (partial sums in float32, reduced deterministically afterwards).
constexpr int kWgTile = 128;

__device__ __forceinline__ uint32_t perm_lo(uint32_t x, uint32_t y) {   // (x.lo, y.lo)
    return __builtin_amdgcn_perm(y, x, 0x05040100u);
}
__device__ __forceinline__ uint32_t perm_hi(uint32_t x, uint32_t y) {   // (x.hi, y.hi)
    return __builtin_amdgcn_perm(y, x, 0x07060302u);
}

// in[p] = channels (c0,c1 | c2,c3) of pixel p (p = 0..7)  ->  out[a] = pixels 0..7 of channel a
__device__ __forceinline__ void transpose8x4(const uint2 (&in)[8], Frag (&out)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint2 x = in[2 * q], y = in[2 * q + 1];
        const uint32_t c0 = perm_lo(x.x, y.x), c1 = perm_hi(x.x, y.x);
        const uint32_t c2 = perm_lo(x.y, y.y), c3 = perm_hi(x.y, y.y);
        uint32_t* o0 = reinterpret_cast<uint32_t*>(&out[0].u);
        uint32_t* o1 = reinterpret_cast<uint32_t*>(&out[1].u);
        uint32_t* o2 = reinterpret_cast<uint32_t*>(&out[2].u);
        uint32_t* o3 = reinterpret_cast<uint32_t*>(&out[3].u);
        o0[q] = c0; o1[q] = c1; o2[q] = c2; o3[q] = c3;
    }
}

__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(ConvDims d, int tiles_ci, int ksteps_per_block,
                                                           const uint16_t* __restrict__ dy,
                                                           const uint16_t* __restrict__ x,
                                                           float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int wa = wave & 1, wb = wave >> 1;                 // co half / ci half of the 128 x 128 tile
    const int tco = blockIdx.x / tiles_ci, tci = blockIdx.x - tco * tiles_ci;
    const int tap = blockIdx.y;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int co_l = tco * kWgTile + 8 * li + 4 * wa;        // this lane's 4 output channels (operand A rows)
    const int ci_l = tci * kWgTile + 8 * li + 4 * wb;        // this lane's 4 input channels  (operand B columns)
    const bool co_ok = co_l < d.Cout;                        // Cout, Cin are multiples of 4 here (checked by the host)
    const bool ci_ok = ci_l < d.Cin;
    const int step0 = blockIdx.z * ksteps_per_block;
    const int total_steps = (d.M + 31) / 32;
    const int step1 = min(step0 + ksteps_per_block, total_steps);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint2 zero2 = make_uint2(0, 0);
    auto load_step = [&](int step, uint2 (&ga)[8], uint2 (&gb)[8]) {
        const int m0 = step * 32 + kq * 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int m = m0 + p;
            const bool mv = m < d.M;
            ga[p] = (mv && co_ok) ? *reinterpret_cast<const uint2*>(dy + (size_t)m * d.ldy + co_l) : zero2;
            const int mm = mv ? m : 0;
            const int wo = mm % d.Wo;
            const int t = mm / d.Wo;
            const int ho = t % d.Ho;
            const int n = t / d.Ho;
            const int hi = ho * d.stride - d.pad_h + kh * d.dil_h;
            const int wi = wo * d.stride - d.pad_w + kw * d.dil_w;
            const bool ok = mv && ci_ok && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
            gb[p] = ok ? *reinterpret_cast<const uint2*>(x + ((size_t)(n * d.H + hi) * d.W + wi) * d.ldx + ci_l) : zero2;
        }
    };

    uint2 ga0[8], gb0[8], ga1[8], gb1[8];
    if (step0 < step1) load_step(step0, ga0, gb0);
    for (int step = step0; step < step1; step += 2) {
        if (step + 1 < step1) load_step(step + 1, ga1, gb1);
        {
            Frag fa[4], fb[4];
            transpose8x4(ga0, fa);
            transpose8x4(gb0, fb);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0);
        }
        if (step + 1 < step1) {
            if (step + 2 < step1) load_step(step + 2, ga0, gb0);
            Frag fa[4], fb[4];
            transpose8x4(ga1, fa);
            transpose8x4(gb1, fb);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0);
        }
    }

    // D[row][col]: row (A index) = (lane>>4)*4 + r -> co = tile + 8*row + 4*wa + a; col = lane&15 -> ci = tile + 8*col + 4*wb + b
    const size_t wsize = (size_t)d.Cout * d.KH * d.KW * d.Cin;
    float* out = partial + (size_t)blockIdx.z * wsize;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = tco * kWgTile + 8 * (kq * 4 + r) + 4 * wa + a;
            if (co >= d.Cout) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ci = tci * kWgTile + 8 * li + 4 * wb + b;
                if (ci < d.Cin) out[((size_t)co * d.KH * d.KW + tap) * d.Cin + ci] = acc[a][b][r];
            }
        }
    }
}

// dw[i] = sum_k partial[k][i]: 16 columns x 16 split lanes per workgroup, fixed tree order (deterministic)
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(int splits, size_t n, const float* __restrict__ partial,
                                                                  float* __restrict__ dw) {
    __shared__ float red[256];
    const int il = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + il;
    float s = 0.f;
    if (i < n)
        for (int k = kl; k < splits; k += 16) s += partial[(size_t)k * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 8; st > 0; st >>= 1) {
        if (kl < st) red[threadIdx.x] += red[threadIdx.x + st * 16];
        __syncthreads();
    }
    if (kl == 0 && i < n) dw[i] = red[il];
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

}  // namespace

extern "C" {

int stp3_conv2d_fwd(const stp3_conv_dims* p, const void* x, const void* w, const float* bias, void* y, void* stream) {
    if (!p || !x || !w || !y) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->KH <= 0 ||
        p->KW <= 0 || p->stride <= 0 || p->dil_h <= 0 || p->dil_w <= 0 || p->pad_h < 0 || p->pad_w < 0)
        return STP3_EINVAL;
    if (p->has_bias && !bias) return STP3_EINVAL;
    if (p->Cin % 8 || p->ldx % 8 || p->ldx < p->Cin || p->ldy < p->Cout) return STP3_EUNSUP;   // 16-byte k-chunks
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return STP3_EUNSUP;
    if (p->out_dtype != STP3_DTYPE_BF16 && p->out_dtype != STP3_DTYPE_F32) return STP3_EUNSUP;
    if (((uintptr_t)y & (p->out_dtype == STP3_DTYPE_F32 ? 15 : 7))) return STP3_EUNSUP;
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    if (M >= (1LL << 31) || (int64_t)p->N * p->H * p->W >= (1LL << 31)) return STP3_EUNSUP;
    ConvDims d;
    d.N = p->N; d.H = p->H; d.W = p->W; d.Cin = p->Cin; d.Ho = p->Ho; d.Wo = p->Wo; d.Cout = p->Cout;
    d.KH = p->KH; d.KW = p->KW; d.stride = p->stride; d.pad_h = p->pad_h; d.pad_w = p->pad_w;
    d.dil_h = p->dil_h; d.dil_w = p->dil_w; d.ldx = p->ldx; d.ldy = p->ldy;
    d.out_f32 = p->out_dtype == STP3_DTYPE_F32; d.has_bias = p->has_bias;
    d.M = (int)M; d.kchunks = (p->Cin + 31) / 32;
    // A/B switch (STP3_CONV_KERNEL=v2): bf16-output convolutions through the v2 kernel of stp3_conv2.hip (16-channel
    // stores, operand loads that stay global: no scratch, 3 waves per SIMD) instead of the kernel below
    static const bool use_v2 = [] {
        const char* e = getenv("STP3_CONV_KERNEL");
        return e && !strcmp(e, "v2");
    }();
    if (use_v2 && p->out_dtype == STP3_DTYPE_BF16 && p->ldx % 8 == 0 && !((uintptr_t)y & 15))
        return stp3_conv2d_fwd_v2(p, x, w, bias, y, nullptr, nullptr, 0, stream);
    dim3 grid((unsigned)((M + kTilePix - 1) / kTilePix), (unsigned)((p->Cout + kTileCo - 1) / kTileCo));
    hipLaunchKernelGGL(conv2d_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, (const uint16_t*)x,
                       (const uint16_t*)w, bias, y);
    return status();
}


static int wgrad_plan(const stp3_conv_dims* p, int* tiles_co, int* tiles_ci, int* splits, int* ksteps) {
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    *tiles_co = (p->Cout + kWgTile - 1) / kWgTile;
    *tiles_ci = (p->Cin + kWgTile - 1) / kWgTile;
    const int64_t total_steps = (M + 31) / 32;
    const int64_t base = (int64_t)(*tiles_co) * (*tiles_ci) * p->KH * p->KW;
    int64_t want = (1024 + base - 1) / base;                 // ~1024 workgroups
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
    int tco, tci, splits, ksteps;
    wgrad_plan(p, &tco, &tci, &splits, &ksteps);
    *bytes = (size_t)splits * p->Cout * p->KH * p->KW * p->Cin * sizeof(float);
    return STP3_OK;
}

int stp3_conv2d_wgrad(const stp3_conv_dims* p, const void* dy, const void* x, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream) {
    if (!p || !dy || !x || !dw || !workspace) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->KH <= 0 ||
        p->KW <= 0 || p->stride <= 0 || p->dil_h <= 0 || p->dil_w <= 0 || p->pad_h < 0 || p->pad_w < 0)
        return STP3_EINVAL;
    // 8-byte channel quads of both operands
    if (p->Cin % 4 || p->Cout % 4 || p->ldx % 4 || p->ldy % 4 || p->ldx < p->Cin || p->ldy < p->Cout) return STP3_EUNSUP;
    if (((uintptr_t)x & 7) || ((uintptr_t)dy & 7)) return STP3_EUNSUP;
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    if (M >= (1LL << 31) - 64 || (int64_t)p->N * p->H * p->W >= (1LL << 31)) return STP3_EUNSUP;
    int tco, tci, splits, ksteps;
    wgrad_plan(p, &tco, &tci, &splits, &ksteps);
    const size_t wsize = (size_t)p->Cout * p->KH * p->KW * p->Cin;
    if (workspace_bytes < (size_t)splits * wsize * sizeof(float)) return STP3_ENOSPACE;
    ConvDims d;
    d.N = p->N; d.H = p->H; d.W = p->W; d.Cin = p->Cin; d.Ho = p->Ho; d.Wo = p->Wo; d.Cout = p->Cout;
    d.KH = p->KH; d.KW = p->KW; d.stride = p->stride; d.pad_h = p->pad_h; d.pad_w = p->pad_w;
    d.dil_h = p->dil_h; d.dil_w = p->dil_w; d.ldx = p->ldx; d.ldy = p->ldy;
    d.out_f32 = 1; d.has_bias = 0; d.M = (int)M; d.kchunks = 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv2d_wgrad_kernel, dim3(tco * tci, p->KH * p->KW, splits), dim3(256), 0, s, d, tci, ksteps,
                       (const uint16_t*)dy, (const uint16_t*)x, (float*)workspace);
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((wsize + 15) / 16)), dim3(256), 0, s, splits, wsize,
                       (const float*)workspace, dw);
    return status();
}

}  // extern "C"
