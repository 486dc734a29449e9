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
    dim3 grid((unsigned)((M + kTilePix - 1) / kTilePix), (unsigned)((p->Cout + kTileCo - 1) / kTileCo));
    hipLaunchKernelGGL(conv2d_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, (const uint16_t*)x,
                       (const uint16_t*)w, bias, y);
    return status();
}

}  // extern "C"
