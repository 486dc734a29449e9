// stp3_conv2.hip -- second-generation forward of the bf16 MFMA implicit-GEMM convolution (stp3_conv.hip):
//   (1) the weight rows a lane feeds to the MFMA are permuted so that an accumulator lane ends up with 16
//       CONSECUTIVE output channels of one pixel: the epilogue writes 32 contiguous bytes per lane and 128
//       contiguous bytes per pixel (v1: 8-byte pieces, 32 bytes per pixel per store instruction) -- what the
//       activation-bandwidth-bound pointwise layers of the EfficientNet trunk need;
//   (2) optional BatchNorm statistics in the epilogue: per-channel sum and sum of squares of the (bf16-rounded)
//       outputs, reduced over the workgroup's 128 pixels in registers / LDS and written as one partial row per
//       workgroup -- the separate statistics pass over the convolution output (stp3_bn_stats) disappears.
// Same reference scope as stp3_conv.hip (nn.Conv2d -> nn.BatchNorm2d chains of stp3/layers/convolutions.py:183-280,
// stp3/layers/temporal.py:252-325, stp3/models/decoder.py:22-140, MBConv 1x1 convolutions).
// STATUS: compiled and exported, selected only with STP3_CONV_V2=1 (host side: stp3_amd/ops_fused.py); not yet
// validated on hardware -- the default path is stp3_conv.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvDims2 {
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int ldx, ldy;
    int has_bias;
    int M, kchunks;
};

union Frag2 {
    uint4 u;
    bf16x8 v;
};

__device__ __forceinline__ uint32_t f2bf2(float a) {   // round to nearest even
    uint32_t u = __float_as_uint(a);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf2f2(uint32_t b) { return __uint_as_float(b << 16); }

template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_row<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_row<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_row<0x141>(v);   // row_half_mirror
    v += dpp_row<0x140>(v);   // row_mirror
    return v;
}

constexpr int kPix = 128, kCo = 64, PT2 = 2, CT2 = 4;

template <bool STATS>
__global__ __launch_bounds__(256) void conv2d_fwd_v2_kernel(ConvDims2 d, const uint16_t* __restrict__ x,
                                                            const uint16_t* __restrict__ w,
                                                            const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                            float* __restrict__ partial) {
    __shared__ float red[4][2][kCo];                     // [wave][sum | sumsq][channel of the block]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int m_wave = blockIdx.x * kPix + wave * (PT2 * 16);
    const int co0 = blockIdx.y * kCo;

    int pn[PT2], ph[PT2], pw[PT2];
    bool pvalid[PT2];
#pragma unroll
    for (int j = 0; j < PT2; ++j) {
        const int m = m_wave + j * 16 + li;
        pvalid[j] = m < d.M;
        const int mm = pvalid[j] ? m : 0;
        const int wo = mm % d.Wo;
        const int t = mm / d.Wo;
        pn[j] = t / d.Ho;
        ph[j] = (t % d.Ho) * d.stride - d.pad_h;
        pw[j] = wo * d.stride - d.pad_w;
    }
    // A-fragment row li of tile i stands for channel co0 + (li>>2)*16 + i*4 + (li&3): the accumulator row
    // (lane>>4)*4 + r of tile i is then channel co0 + kq*16 + i*4 + r -- 16 consecutive channels per lane.
    const size_t wrow = (size_t)d.KH * d.KW * d.Cin;
    bool cvalid[CT2];
    const uint16_t* wp[CT2];
#pragma unroll
    for (int i = 0; i < CT2; ++i) {
        const int co = co0 + (li >> 2) * 16 + i * 4 + (li & 3);
        cvalid[i] = co < d.Cout;
        wp[i] = w + (size_t)(cvalid[i] ? co : 0) * wrow + kq * 8;
    }

    f32x4 acc[CT2][PT2];
#pragma unroll
    for (int i = 0; i < CT2; ++i)
#pragma unroll
        for (int j = 0; j < PT2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int steps = d.KH * d.KW * d.kchunks;
    auto load_step = [&](int step, Frag2 (&a)[CT2], Frag2 (&b)[PT2]) {
        const int tap = step / d.kchunks;
        const int cbase = (step - tap * d.kchunks) * 32;
        const int c0 = cbase + kq * 8;
        const bool kvalid = c0 < d.Cin;
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        // Out-of-range fragments load from a valid GLOBAL address (the tensor's first element) and are zeroed in
        // registers, component by component.  Written as `ok ? *p : zero4` the compiler selects between the global
        // address and the address of a zero constant in scratch, which turns every operand load into a flat_load
        // (both wait counters, 32 bytes of scratch); v1 (stp3_conv.hip) still has that form.
#pragma unroll
        for (int i = 0; i < CT2; ++i) {
            const bool ok = cvalid[i] && kvalid;
            const uint4 v = *reinterpret_cast<const uint4*>(ok ? wp[i] + (size_t)tap * d.Cin + cbase : w);
            a[i].u = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
#pragma unroll
        for (int j = 0; j < PT2; ++j) {
            const int hi = ph[j] + kh * d.dil_h;
            const int wi = pw[j] + kw * d.dil_w;
            const bool ok = pvalid[j] && kvalid && hi >= 0 && hi < d.H && wi >= 0 && wi < d.W;
            const uint4 v = *reinterpret_cast<const uint4*>(ok ? x + ((size_t)(pn[j] * d.H + hi) * d.W + wi) * d.ldx + c0 : x);
            b[j].u = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
    };

    Frag2 a0[CT2], b0[PT2], a1[CT2], b1[PT2];
    load_step(0, a0, b0);
    for (int step = 0; step < steps; step += 2) {
        if (step + 1 < steps) load_step(step + 1, a1, b1);
#pragma unroll
        for (int i = 0; i < CT2; ++i)
#pragma unroll
            for (int j = 0; j < PT2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i].v, b0[j].v, acc[i][j], 0, 0, 0);
        if (step + 1 < steps) {
            if (step + 2 < steps) load_step(step + 2, a0, b0);
#pragma unroll
            for (int i = 0; i < CT2; ++i)
#pragma unroll
                for (int j = 0; j < PT2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i].v, b1[j].v, acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane (pixel li, quarter kq) holds channels cb .. cb+15, element e = i*4 + r --------------
    const int cb = co0 + kq * 16;
    float bsum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bsum[e] = (d.has_bias && cb + e < d.Cout) ? bias[cb + e] : 0.f;
    float s[16], q[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = q[e] = 0.f;
    const bool vec = cb + 15 < d.Cout && (d.ldy & 7) == 0;
#pragma unroll
    for (int j = 0; j < PT2; ++j) {
        const int m = m_wave + j * 16 + li;
        const bool mv = m < d.M;
        uint32_t hb[16];
#pragma unroll
        for (int i = 0; i < CT2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) hb[i * 4 + r] = f2bf2(acc[i][j][r] + bsum[i * 4 + r]);
        if (STATS && mv) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = bf2f2(hb[e]);                // statistics of what BatchNorm will read
                s[e] += v;
                q[e] = fmaf(v, v, q[e]);
            }
        }
        if (mv && cb < d.Cout) {
            uint16_t* yp = y + (size_t)m * d.ldy + cb;
            if (vec) {
                uint4 lo = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
                uint4 hi = make_uint4(hb[8] | (hb[9] << 16), hb[10] | (hb[11] << 16), hb[12] | (hb[13] << 16),
                                      hb[14] | (hb[15] << 16));
                *reinterpret_cast<uint4*>(yp) = lo;
                *reinterpret_cast<uint4*>(yp + 8) = hi;
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (cb + e < d.Cout) yp[e] = (uint16_t)hb[e];
            }
        }
    }
    if (STATS) {
        // sum over the 16 pixels a DPP row holds (lanes of one kq), then over the 4 waves through LDS
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = row_sum16(s[e]);
            q[e] = row_sum16(q[e]);
        }
        if (li == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                red[wave][0][kq * 16 + e] = s[e];
                red[wave][1][kq * 16 + e] = q[e];
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * kCo) {
            const int k = threadIdx.x / kCo, c = threadIdx.x - k * kCo;
            const float t = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
            if (co0 + c < d.Cout) partial[((size_t)blockIdx.x * 2 + k) * d.Cout + co0 + c] = t;
        }
    }
}

// out[i] = sum_p partial[p][i] (double accumulation), 16 columns x 16 row lanes per workgroup
__global__ __launch_bounds__(256) void colsum_kernel(int parts, int width, const float* __restrict__ partial,
                                                     float* __restrict__ out) {
    __shared__ double red[256];
    const int il = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    double s = 0.0;
    if (i < width)
        for (int p = pl; p < parts; p += 16) s += (double)partial[(size_t)p * width + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 8; st > 0; st >>= 1) {
        if (pl < st) red[threadIdx.x] += red[threadIdx.x + st * 16];
        __syncthreads();
    }
    if (pl == 0 && i < width) out[i] = (float)red[il];
}

inline int status2() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

}  // namespace

extern "C" {

int stp3_conv2d_fwd_v2_workspace(const stp3_conv_dims* p, size_t* bytes) {
    if (!p || !bytes || p->N <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->Cout <= 0) return STP3_EINVAL;
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    *bytes = (size_t)((M + kPix - 1) / kPix) * 2 * p->Cout * sizeof(float);
    return STP3_OK;
}

int stp3_conv2d_fwd_v2(const stp3_conv_dims* p, const void* x, const void* w, const float* bias, void* y, float* sums,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !x || !w || !y) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->Ho <= 0 || p->Wo <= 0 || p->KH <= 0 ||
        p->KW <= 0 || p->stride <= 0 || p->dil_h <= 0 || p->dil_w <= 0 || p->pad_h < 0 || p->pad_w < 0)
        return STP3_EINVAL;
    if (p->has_bias && !bias) return STP3_EINVAL;
    if (p->out_dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (p->Cin % 8 || p->ldx % 8 || p->ldx < p->Cin || p->ldy < p->Cout) return STP3_EUNSUP;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return STP3_EUNSUP;
    const int64_t M = (int64_t)p->N * p->Ho * p->Wo;
    if (M >= (1LL << 31) || (int64_t)p->N * p->H * p->W >= (1LL << 31)) return STP3_EUNSUP;
    ConvDims2 d;
    d.N = p->N; d.H = p->H; d.W = p->W; d.Cin = p->Cin; d.Ho = p->Ho; d.Wo = p->Wo; d.Cout = p->Cout;
    d.KH = p->KH; d.KW = p->KW; d.stride = p->stride; d.pad_h = p->pad_h; d.pad_w = p->pad_w;
    d.dil_h = p->dil_h; d.dil_w = p->dil_w; d.ldx = p->ldx; d.ldy = p->ldy; d.has_bias = p->has_bias;
    d.M = (int)M; d.kchunks = (p->Cin + 31) / 32;
    const unsigned gx = (unsigned)((M + kPix - 1) / kPix);
    dim3 grid(gx, (unsigned)((p->Cout + kCo - 1) / kCo));
    hipStream_t s = (hipStream_t)stream;
    if (sums) {
        if (!workspace || workspace_bytes < (size_t)gx * 2 * p->Cout * sizeof(float)) return STP3_ENOSPACE;
        hipLaunchKernelGGL((conv2d_fwd_v2_kernel<true>), grid, dim3(256), 0, s, d, (const uint16_t*)x, (const uint16_t*)w,
                           bias, (uint16_t*)y, (float*)workspace);
        hipLaunchKernelGGL(colsum_kernel, dim3((2 * p->Cout + 15) / 16), dim3(256), 0, s, (int)gx, 2 * p->Cout,
                           (const float*)workspace, sums);
    } else {
        hipLaunchKernelGGL((conv2d_fwd_v2_kernel<false>), grid, dim3(256), 0, s, d, (const uint16_t*)x, (const uint16_t*)w,
                           bias, (uint16_t*)y, (float*)nullptr);
    }
    return status2();
}

}  // extern "C"
