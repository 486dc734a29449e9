// stp3_dwconv.hip -- depthwise 2-D convolution (forward, data gradient, weight gradient) for the
// EfficientNet trunk of ST-P3's image encoder on gfx950.
//
// Reference: the trunk's MBConv blocks call efficientnet_pytorch's Conv2dStaticSamePadding with
// groups == channels (reference stp3/models/encoder.py:62-70 drives them; the package itself is
// not vendored).  These layers are HBM-bound (k*k MACs per element): the kernels below are
// channels-last (NHWC), vectorised 16 B per lane along C, fp32 accumulation, bf16 or f32 I/O.
// MIOpen's grouped-conv path for this shape costs ~48 ms per weight-gradient call on MI355X
// (profiles/r01_bench_steady_miopen.txt); these kernels replace it.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stdlib.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

struct DwDims {
    int N, H, W, C, Ho, Wo;
    int pad_t, pad_l;
};

// ---- 16-byte vectors of 8 bf16 / 4 f32 -------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    __device__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ void to_float(float* f) const { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
    __device__ void from_float(const float* f) { v = make_float4(f[0], f[1], f[2], f[3]); }
};
struct bf16x8 { uint4 v; };
template <> struct Vec<uint16_t> {
    static constexpr int N = 8;
    uint4 v;
    __device__ void load(const uint16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ void store(uint16_t* p) const { *reinterpret_cast<uint4*>(p) = v; }
    __device__ void zero() { v = make_uint4(0, 0, 0, 0); }
    __device__ void to_float(float* f) const {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ void from_float(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16(f[2 * i], f[2 * i + 1]);
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// Addressing.  The kernels below are instruction-bound on the 5x5 / 7x7 layers (25 / 49 multiply-adds per output against 2 + 2
// bytes; profiles/r05x_step_pmc.json: 8-11 VALU lane-instructions per byte), and with 64-bit element offsets a third of those
// instructions was address arithmetic: a quarter-rate v_mad_u64_u32 per 16-byte load, 64-bit divisions in every thread's
// prologue.  `Off` is the type of a BYTE offset from the (uniform) tensor base: uint32_t whenever every tensor of the call is
// smaller than 4 GiB -- the loads then take the scalar-base + 32-bit-offset form --, uint64_t otherwise.
template <typename T, typename Off>
__device__ __forceinline__ const T* at(const T* base, Off byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T, typename Off>
__device__ __forceinline__ T* at(T* base, Off byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}

// XCD-aware workgroup order (see stp3_conv.hip): the dispatcher places workgroup b on XCD b % 8, each XCD has its own L2, and
// neighbouring workgroups read overlapping input rows (every input row serves K output rows).  v = xcd_order(b, n): XCD x
// runs the x-th contiguous chunk of the launch order, so the K - 1 halo rows are L2 hits instead of K separate fills.
__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- forward: y[n,ho,wo,c] = sum_{kh,kw} x[n, ho*S+kh-pt, wo*S+kw-pl, c] * w[kh,kw,c] ----------
// One thread = one channel vector x TW consecutive output columns: the input row segment and the
// K weight vectors of a kernel row are loaded once and reused across the TW outputs.
// FLIP: the taps are read in reverse order -- the data gradient of a stride-1 depthwise convolution IS a depthwise
// convolution of dy with the flipped kernel (padding K - 1 - p), so it shares this kernel and its register reuse
// (TW outputs per thread share one input span per kernel row; the tap-major weights are read once per row).
template <typename T, int K, int S, int TW, bool FLIP, typename Off>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwDims d, const T* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y) {
    constexpr int VN = Vec<T>::N;
    const int CV = d.C / VN;
    const int wgroups = (d.Wo + TW - 1) / TW;
    const Off total = (Off)d.N * d.Ho * wgroups * CV;
    const Off tid = (Off)xcd_order(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (tid >= total) return;
    const int cv = (int)(tid % (Off)CV);
    Off r = tid / (Off)CV;
    const int wg = (int)(r % (Off)wgroups);
    r /= (Off)wgroups;
    const int ho = (int)(r % (Off)d.Ho);
    const int n = (int)(r / (Off)d.Ho);
    const int c0 = cv * VN;
    const int wo0 = wg * TW;
    float acc[TW][VN];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < VN; ++j) acc[i][j] = bias ? bias[c0 + j] : 0.f;     // (the ConvNeXt blocks' 7x7 layers have one)
    constexpr int SPAN = (TW - 1) * S + K;
    const int wi0 = wo0 * S - d.pad_l;
    const Off pix = (Off)d.C * sizeof(T);                     // bytes from one pixel to the next
    const Off wtap = (Off)d.C * sizeof(float);                // bytes from one tap's weight row to the next
    const Off wc0 = (Off)c0 * sizeof(float);
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int hi = ho * S + kh - d.pad_t;
        if (hi < 0 || hi >= d.H) continue;
        float wk[K][VN];
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const float* wp = at(w, (Off)(FLIP ? (K - 1 - kh) * K + (K - 1 - kw) : kh * K + kw) * wtap + wc0);
#pragma unroll
            for (int j = 0; j < VN; ++j) wk[kw][j] = wp[j];
        }
        const Off xrow = ((Off)(n * d.H + hi) * d.W) * pix + (Off)c0 * sizeof(T);
        float xin[SPAN][VN];
#pragma unroll
        for (int i = 0; i < SPAN; ++i) {
            const int wi = wi0 + i;
            Vec<T> v;
            if (wi >= 0 && wi < d.W) v.load(at(x, xrow + (Off)wi * pix)); else v.zero();
            v.to_float(xin[i]);
        }
#pragma unroll
        for (int o = 0; o < TW; ++o)
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int j = 0; j < VN; ++j) acc[o][j] = fmaf(xin[o * S + kw][j], wk[kw][j], acc[o][j]);
    }
    const Off yrow = ((Off)(n * d.Ho + ho) * d.Wo) * pix + (Off)c0 * sizeof(T);
#pragma unroll
    for (int o = 0; o < TW; ++o) {
        if (wo0 + o < d.Wo) {
            Vec<T> v;
            v.from_float(acc[o]);
            v.store(at(y, yrow + (Off)(wo0 + o) * pix));
        }
    }
}

// ---- data gradient: dx[n,hi,wi,c] = sum_{kh,kw : (hi+pt-kh) = ho*S, (wi+pl-kw) = wo*S} dy[n,ho,wo,c] w[kh,kw,c]
// Stride 1: the forward kernel on dy with flipped taps (launch_bwd_data).  STRIDE 2, one thread per 2 x 2 input quad.
// With a thread per input pixel (rounds 1-3) the taps that reach a pixel depend on the parity of its row and column
// (1, 2 or 4 of the 9 taps of a 3x3 kernel), so the lanes of a wave -- a few neighbouring pixels x all channel vectors --
// walked through all nine tap branches under masks, each with its own gradient and weight loads (27 vector loads compiled,
// 1.7 TB/s on the 532-MiB gradient of the first stride-2 block).  A 2 x 2 quad of input pixels (rows 2a - pt, 2a + 1 - pt)
// sees every tap exactly once and reads the same R x R gradient pixels (R = (K + 1) / 2: rows a - R + 1 .. a): no
// divergence, R * R gradient loads for four outputs.  Persistent threads with a fixed channel vector: a 3x3 kernel keeps its
// 9 x 8 weights in registers (5x5: read per tap, L1-resident).
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_data_s2_kernel(DwDims d, int qrows, int qcols, const T* __restrict__ dy,
                                                                 const float* __restrict__ w, T* __restrict__ dx) {
    constexpr int VN = Vec<T>::N;
    constexpr int R = (K + 1) / 2;
    constexpr bool kRegW = K == 3;
    const int CV = d.C / VN;
    const int CVB = min(CV, 256);
    const int QL = 256 / CVB;                         // quads per workgroup pass
    const int cv = blockIdx.y * CVB + threadIdx.x % CVB, ql = threadIdx.x / CVB;
    if (ql >= QL || cv >= CV) return;
    const int c0 = cv * VN;
    float wr[kRegW ? K * K : 1][VN];
    if (kRegW) {
#pragma unroll
        for (int t = 0; t < K * K; ++t)
#pragma unroll
            for (int j = 0; j < VN; ++j) wr[t][j] = w[t * d.C + c0 + j];
    }
    const int a0 = d.pad_t >> 1, b0 = d.pad_l >> 1;   // quad row / column of input row / column 0
    const int64_t total = (int64_t)d.N * qrows * qcols;
    for (int64_t q = (int64_t)blockIdx.x * QL + ql; q < total; q += (int64_t)gridDim.x * QL) {
        const int bi = (int)(q % qcols);
        int64_t r = q / qcols;
        const int ai = (int)(r % qrows);
        const int n = (int)(r / qrows);
        const int a = a0 + ai, b = b0 + bi;
        float acc[2][2][VN];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int j = 0; j < VN; ++j) acc[pr][pc][j] = 0.f;
#pragma unroll
        for (int dr = 0; dr < R; ++dr) {
            const int ho = a - dr;
            if (ho < 0 || ho >= d.Ho) continue;
#pragma unroll
            for (int dc = 0; dc < R; ++dc) {
                const int wo = b - dc;
                if (wo < 0 || wo >= d.Wo) continue;
                Vec<T> v;
                v.load(dy + ((int64_t)(n * d.Ho + ho) * d.Wo + wo) * d.C + c0);
                float g[VN];
                v.to_float(g);
                // input row 2a + pr - pt takes the taps kh = 2 dr + pr, column 2b + pc - pl the taps kw = 2 dc + pc
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int kh = 2 * dr + pr;
                    if (kh >= K) continue;
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int kw = 2 * dc + pc;
                        if (kw >= K) continue;
                        if (kRegW) {
#pragma unroll
                            for (int j = 0; j < VN; ++j) acc[pr][pc][j] = fmaf(g[j], wr[kh * K + kw][j], acc[pr][pc][j]);
                        } else {
#pragma unroll
                            for (int j = 0; j < VN; ++j)
                                acc[pr][pc][j] = fmaf(g[j], w[(kh * K + kw) * d.C + c0 + j], acc[pr][pc][j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int hi = 2 * a + pr - d.pad_t;
            if (hi < 0 || hi >= d.H) continue;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int wi = 2 * b + pc - d.pad_l;
                if (wi < 0 || wi >= d.W) continue;
                Vec<T> o;
                o.from_float(acc[pr][pc]);
                o.store(dx + ((int64_t)(n * d.H + hi) * d.W + wi) * d.C + c0);
            }
        }
    }
}

// ---- weight gradient: dw[kh,kw,c] = sum_{n,ho,wo} dy[n,ho,wo,c] * x[n,ho*S+kh-pt,wo*S+kw-pl,c] ---
// Stage 1: block (bx, by, kh) reduces its strided share of the output pixels for one kernel ROW kh
// into partial[bx][kh*K + kw][C] (threads = channel vectors x pixel lanes, coalesced along C; the
// pixel lanes are summed through LDS in a fixed order).  Stage 2 sums the partials over bx.
// No atomics: deterministic.
template <typename T, int K, int S, typename Off>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(DwDims d, int nbx, const T* __restrict__ x,
                                                                const T* __restrict__ dy,
                                                                float* __restrict__ partial) {
    constexpr int VN = Vec<T>::N;
    constexpr int KK = K * K;
    constexpr int TW = 4;                            // consecutive output pixels of a row per thread and step:
    constexpr int SPAN = (TW - 1) * S + K;           // they share one input span (SPAN loads for TW * K products)
    extern __shared__ __attribute__((aligned(16))) float red[];   // [PL][K][CVB*VN]
    const int CV = d.C / VN;
    const int CVB = min(CV, 256);                    // channel vectors handled by this block column
    const int PL = 256 / CVB;                        // pixel lanes
    const int cvb = threadIdx.x % CVB, pl = threadIdx.x / CVB;
    // 1-D launch, kernel row fastest, XCD-contiguous: the K workgroups that walk the SAME pixels for the K kernel rows sit
    // next to each other on one XCD, so dy (read by all K) and the input rows (each needed by K (output row, kernel row)
    // pairs) come out of that XCD's L2 instead of being fetched K times from HBM
    const int v = xcd_order(blockIdx.x, gridDim.x);
    const int kh = v % K;
    const int bxi = (v / K) % nbx, byi = v / (K * nbx);
    const int cv = byi * CVB + cvb;
    const bool live = pl < PL && cv < CV;
    const int c0 = cv * VN;
    float acc[K][VN];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int j = 0; j < VN; ++j) acc[t][j] = 0.f;
    const int wgroups = (d.Wo + TW - 1) / TW;
    const int ngroups = d.N * d.Ho * wgroups;        // < 2^31 (checked by the launcher)
    if (live) {
        for (int p = bxi * PL + pl; p < ngroups; p += nbx * PL) {
            const int wg = p % wgroups;
            const int r = p / wgroups;
            const int ho = r % d.Ho;
            const int n = r / d.Ho;
            const int hi = ho * S + kh - d.pad_t;
            if (hi < 0 || hi >= d.H) continue;
            const int wo0 = wg * TW;
            const Off pix = (Off)d.C * sizeof(T);
            const Off grow = ((Off)(n * d.Ho + ho) * d.Wo + wo0) * pix + (Off)c0 * sizeof(T);
            const Off xrow = ((Off)(n * d.H + hi) * d.W) * pix + (Off)c0 * sizeof(T);
            const int wi0 = wo0 * S - d.pad_l;
            float g[TW][VN], xin[SPAN][VN];
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                Vec<T> v;
                if (wo0 + o < d.Wo) v.load(at(dy, grow + (Off)o * pix)); else v.zero();
                v.to_float(g[o]);
            }
#pragma unroll
            for (int i = 0; i < SPAN; ++i) {
                const int wi = wi0 + i;
                Vec<T> v;
                if (wi >= 0 && wi < d.W) v.load(at(x, xrow + (Off)wi * pix)); else v.zero();
                v.to_float(xin[i]);
            }
#pragma unroll
            for (int o = 0; o < TW; ++o)
#pragma unroll
                for (int kw = 0; kw < K; ++kw)
#pragma unroll
                    for (int j = 0; j < VN; ++j) acc[kw][j] = fmaf(g[o][j], xin[o * S + kw][j], acc[kw][j]);
        }
    }
    const int rowlen = CVB * VN;
    if (pl < PL) {
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int j = 0; j < VN; ++j) red[(pl * K + t) * rowlen + cvb * VN + j] = acc[t][j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * rowlen; i += 256) {
        const int t = i / rowlen, cc = i % rowlen;
        const int c = byi * rowlen + cc;
        if (c >= d.C) continue;
        float s = 0.f;
        for (int q = 0; q < PL; ++q) s += red[(q * K + t) * rowlen + cc];
        partial[((int64_t)bxi * KK + kh * K + t) * d.C + c] = s;
    }
}

// dw[i] = sum_b partial[b][i]: 64 columns x 4 block lanes per workgroup (256-byte coalesced row segments, four loads
// in flight per thread), fixed summation order (deterministic)
// channels > 0: the result goes out in the PARAMETER's layout [C][K*K] (i = tap * C + c  ->  c * taps + tap) instead of the
// tap-major [K*K][C] the kernels read their weights in: autograd then takes the tensor as the parameter's gradient as it is
__global__ __launch_bounds__(256) void dwconv_reduce_partials_kernel(int nblocks, int n, const float* __restrict__ partial,
                                                                     float* __restrict__ dw, int channels) {
    __shared__ float red[256];
    const int il = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int b = bl;
        for (; b + 12 < nblocks; b += 16) {
            const float a = partial[(int64_t)b * n + i], c = partial[(int64_t)(b + 4) * n + i];
            const float e = partial[(int64_t)(b + 8) * n + i], f = partial[(int64_t)(b + 12) * n + i];
            s0 += a; s1 += c; s2 += e; s3 += f;
        }
        for (; b < nblocks; b += 4) s0 += partial[(int64_t)b * n + i];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (bl == 0 && i < n) {
        const int o = channels > 0 ? (i % channels) * (n / channels) + i / channels : i;
        dw[o] = (red[il] + red[64 + il]) + (red[128 + il] + red[192 + il]);
    }
}


// ---- forward + BatchNorm statistics of the result -------------------------------------------------------------------
// The forward kernel with a workgroup mapping made for reducing: CVB channel vectors x PL pixel lanes, the workgroup
// walks the output pixels grid-stride, so that a thread keeps ONE channel vector and can carry that vector's sum and
// sum of squares (of the ROUNDED outputs: what a statistics pass over y would read) in registers; one partial row
// [2][C] per workgroup, reduced deterministically by dwconv_stat_reduce_kernel.  Saves the stp3_bn_stats pass over y.
template <typename T, int K, int S, typename Off>
__global__ __launch_bounds__(256) void dwconv_fwd_stats_kernel(DwDims d, const T* __restrict__ x, const float* __restrict__ w,
                                                               T* __restrict__ y, float* __restrict__ partial) {
    constexpr int VN = Vec<T>::N;
    constexpr int TW = 4;
    constexpr int SPAN = (TW - 1) * S + K;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [PL][2][CVB * VN]
    const int CV = d.C / VN;
    const int CVB = min(CV, 256);
    const int PL = 256 / CVB;
    const int cvb = threadIdx.x % CVB, pl = threadIdx.x / CVB;
    const int cv = blockIdx.y * CVB + cvb;
    const bool live = pl < PL && cv < CV;
    const int c0 = cv * VN;
    float s1[VN], s2[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) s1[j] = s2[j] = 0.f;
    const int wgroups = (d.Wo + TW - 1) / TW;
    const int ngroups = d.N * d.Ho * wgroups;            // < 2^31 (checked by the launcher)
    // (grid-stride over the pixel groups; contiguous runs per workgroup in XCD order measured 3-6 % slower here)
    if (live) {
        for (int p = blockIdx.x * PL + pl; p < ngroups; p += gridDim.x * PL) {
            const int wg = p % wgroups;
            const int r = p / wgroups;
            const int ho = r % d.Ho;
            const int n = r / d.Ho;
            const int wo0 = wg * TW;
            float acc[TW][VN];
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < VN; ++j) acc[i][j] = 0.f;
            const int wi0 = wo0 * S - d.pad_l;
            const Off pix = (Off)d.C * sizeof(T);
            const Off wtap = (Off)d.C * sizeof(float), wc0 = (Off)c0 * sizeof(float);
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const int hi = ho * S + kh - d.pad_t;
                if (hi < 0 || hi >= d.H) continue;
                float wk[K][VN];
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const float* wp = at(w, (Off)(kh * K + kw) * wtap + wc0);
#pragma unroll
                    for (int j = 0; j < VN; ++j) wk[kw][j] = wp[j];
                }
                const Off xrow = ((Off)(n * d.H + hi) * d.W) * pix + (Off)c0 * sizeof(T);
                float xin[SPAN][VN];
#pragma unroll
                for (int i = 0; i < SPAN; ++i) {
                    const int wi = wi0 + i;
                    Vec<T> v;
                    if (wi >= 0 && wi < d.W) v.load(at(x, xrow + (Off)wi * pix)); else v.zero();
                    v.to_float(xin[i]);
                }
#pragma unroll
                for (int o = 0; o < TW; ++o)
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int j = 0; j < VN; ++j) acc[o][j] = fmaf(xin[o * S + kw][j], wk[kw][j], acc[o][j]);
            }
            const Off yrow = ((Off)(n * d.Ho + ho) * d.Wo) * pix + (Off)c0 * sizeof(T);
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                if (wo0 + o < d.Wo) {
                    Vec<T> v;
                    v.from_float(acc[o]);
                    v.store(at(y, yrow + (Off)(wo0 + o) * pix));
                    float rv[VN];
                    v.to_float(rv);                         // the stored (rounded) values
#pragma unroll
                    for (int j = 0; j < VN; ++j) {
                        s1[j] += rv[j];
                        s2[j] = fmaf(rv[j], rv[j], s2[j]);
                    }
                }
            }
        }
    }
    const int rowlen = CVB * VN;
    if (pl < PL) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            red[(pl * 2) * rowlen + cvb * VN + j] = live ? s1[j] : 0.f;
            red[(pl * 2 + 1) * rowlen + cvb * VN + j] = live ? s2[j] : 0.f;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * rowlen; i += 256) {
        const int k = i / rowlen, cc = i - k * rowlen;
        const int c = blockIdx.y * rowlen + cc;
        if (c >= d.C) continue;
        float t = 0.f;
        for (int q = 0; q < PL; ++q) t += red[(q * 2 + k) * rowlen + cc];       // pixel lanes in ascending order
        partial[((int64_t)blockIdx.x * 2 + k) * d.C + c] = t;
    }
}

// sums[i] = sum_b partial[b][i] in double, fixed order: 8 columns x 32 block lanes per workgroup, four loads in flight
// per thread (the launch sits between the depthwise pass and everything that needs the statistics: its latency is on the
// critical path, and with up to 2048 partial rows a 64-column workgroup of 4 lanes took 44 us on the MI355X)
constexpr int kRedCols = 8, kRedLanes = 256 / kRedCols;
__global__ __launch_bounds__(256) void dwconv_stat_reduce_kernel(int nblocks, int n, const float* __restrict__ partial,
                                                                 float* __restrict__ sums) {
    __shared__ double red[256];
    const int il = threadIdx.x % kRedCols, bl = threadIdx.x / kRedCols;
    const int i = blockIdx.x * kRedCols + il;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        const float* src = partial + i;
        int b = bl;
        for (; b + 3 * kRedLanes < nblocks; b += 4 * kRedLanes) {
            const float a = src[(int64_t)b * n], c = src[(int64_t)(b + kRedLanes) * n];
            const float e = src[(int64_t)(b + 2 * kRedLanes) * n], f = src[(int64_t)(b + 3 * kRedLanes) * n];
            s0 += (double)a; s1 += (double)c; s2 += (double)e; s3 += (double)f;
        }
        for (; b < nblocks; b += kRedLanes) s0 += (double)src[(int64_t)b * n];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int st = kRedLanes / 2; st > 0; st >>= 1) {
        if (bl < st) red[threadIdx.x] += red[threadIdx.x + st * kRedCols];
        __syncthreads();
    }
    if (bl == 0 && i < n) sums[i] = (float)red[il];
}

// The same sums AND what stp3_bn_finalize makes of them, in one launch (single process: nothing happens between the two).  A
// workgroup owns 4 channels -- its 8 columns are (sum, sum of squares) of those channels, each column added in the order of
// dwconv_stat_reduce_kernel (same bits) -- and its first four threads finish them with bn_finalize_kernel's arithmetic
// (stp3_mbconv.hip: same bits again): coef = scale | shift | mean | invstd, running statistics updated.
struct BnFin {
    float inv_count, unbias, eps, momentum;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* coef;
};
__global__ __launch_bounds__(256) void dwconv_stat_reduce_finalize_kernel(int nblocks, int C, const float* __restrict__ partial,
                                                                          float* __restrict__ sums, BnFin f) {
    __shared__ double red[256];
    const int il = threadIdx.x % kRedCols, bl = threadIdx.x / kRedCols;
    const int c = blockIdx.x * (kRedCols / 2) + (il & 3), k = il >> 2;
    const int n = 2 * C;
    const int i = k * C + c;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < C) {
        const float* src = partial + i;
        int b = bl;
        for (; b + 3 * kRedLanes < nblocks; b += 4 * kRedLanes) {
            const float a = src[(int64_t)b * n], cc = src[(int64_t)(b + kRedLanes) * n];
            const float e = src[(int64_t)(b + 2 * kRedLanes) * n], g = src[(int64_t)(b + 3 * kRedLanes) * n];
            s0 += (double)a; s1 += (double)cc; s2 += (double)e; s3 += (double)g;
        }
        for (; b < nblocks; b += kRedLanes) s0 += (double)src[(int64_t)b * n];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int st = kRedLanes / 2; st > 0; st >>= 1) {
        if (bl < st) red[threadIdx.x] += red[threadIdx.x + st * kRedCols];
        __syncthreads();
    }
    if (bl == 0 && c < C) sums[i] = (float)red[il];
    if (threadIdx.x < 4 && c < C) {                       // il = channel slot, k = 0: both columns of the channel are in red[]
        const float sum = (float)red[il], sq = (float)red[4 + il];
        const float mean = sum * f.inv_count;
        const float var = fmaxf(sq * f.inv_count - mean * mean, 0.f);
        const float invstd = 1.0f / sqrtf(var + f.eps);
        const float scale = (f.gamma ? f.gamma[c] : 1.f) * invstd;
        f.coef[c] = scale;
        f.coef[C + c] = (f.beta ? f.beta[c] : 0.f) - mean * scale;
        f.coef[2 * C + c] = mean;
        f.coef[3 * C + c] = invstd;
        if (f.running_mean) {
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * var * f.unbias;
        }
    }
}

constexpr int kStatBlocks = 2048;
constexpr int kWgradBlocks = 512;

// Workgroups of `kernel` (256 threads, `lds` bytes of dynamic LDS) that the chip's 256 CUs keep resident at once.  The
// persistent kernels below take AT MOST one such round: their blocks walk equal strided shares of the pixels, so blocks beyond
// a round run alone behind it (2048 blocks of dwconv_fwd_stats<3, 1> on 768 resident places were 2.67 rounds).
// The occupancy query costs a runtime call: the answer is cached per (kernel, LDS size) -- the step asks ~60 times -- and
// the CU count is read from the device once.
template <typename Kern>
inline int resident_blocks(Kern kernel, size_t lds) {
    struct Entry { const void* fn; size_t lds; int blocks; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    static int cus = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : cache)
        if (e.fn == (const void*)kernel && e.lds == lds) return e.blocks;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    cache.push_back(Entry{(const void*)kernel, lds, per_cu * cus});
    return per_cu * cus;
}

inline int status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

inline int check(const stp3_dwconv_dims* p, DwDims* d, int* vec) {
    if (!p) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->C <= 0 || p->Ho <= 0 || p->Wo <= 0) return STP3_EINVAL;
    // 3x3 / 5x5 at strides 1 and 2 (EfficientNet trunk), 7x7 at stride 1 (ConvNeXt blocks of the prediction stage)
    if (!(((p->K == 3 || p->K == 5) && (p->stride == 1 || p->stride == 2)) || (p->K == 7 && p->stride == 1)))
        return STP3_EUNSUP;
    if (p->dtype != STP3_DTYPE_F32 && p->dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    *vec = p->dtype == STP3_DTYPE_BF16 ? 8 : 4;
    if (p->C % *vec) return STP3_EUNSUP;
    if ((int64_t)p->N * p->H * p->W * p->C >= (1LL << 40)) return STP3_EUNSUP;
    d->N = p->N; d->H = p->H; d->W = p->W; d->C = p->C; d->Ho = p->Ho; d->Wo = p->Wo;
    d->pad_t = p->pad_top; d->pad_l = p->pad_left;
    return STP3_OK;
}

// every tensor of the call (input, output, tap-major weights) below 4 GiB, with a pixel row of slack for the offsets of
// padded columns that are formed but never dereferenced: byte offsets fit 32 bits
template <typename T>
inline bool small_tensors(const DwDims& d) {
    const uint64_t lim = (1ull << 32) - (uint64_t)(d.W + d.Wo + 16) * d.C * sizeof(float);
    return (uint64_t)d.N * d.H * d.W * d.C * sizeof(T) < lim && (uint64_t)d.N * d.Ho * d.Wo * d.C * sizeof(T) < lim &&
           (uint64_t)64 * d.C * sizeof(float) < lim;
}

template <typename T, int K, int S>
int launch_fwd(const DwDims& d, const void* x, const float* w, const float* bias, void* y, hipStream_t s) {
    constexpr int TW = 4;
    const int CV = d.C / Vec<T>::N;
    const int64_t total = (int64_t)d.N * d.Ho * ((d.Wo + TW - 1) / TW) * CV;
    if (small_tensors<T>(d))
        hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, S, TW, false, uint32_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           s, d, (const T*)x, w, bias, (T*)y);
    else
        hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, S, TW, false, uint64_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           s, d, (const T*)x, w, bias, (T*)y);
    return status();
}
template <typename T, int K, int S>
int launch_fwd_stats(const DwDims& d, const void* x, const float* w, void* y, float* sums, float* ws, hipStream_t s,
                     const BnFin* fin = nullptr) {
    constexpr int VN = Vec<T>::N;
    const int CV = d.C / VN;
    const int CVB = CV < 256 ? CV : 256;
    const int PL = 256 / CVB;
    const int64_t ngroups = (int64_t)d.N * d.Ho * ((d.Wo + 3) / 4);
    if (ngroups >= (1LL << 31)) return STP3_EUNSUP;
    const int by = (CV + CVB - 1) / CVB;
    int64_t want = (ngroups + PL - 1) / PL;
    const size_t lds = (size_t)PL * 2 * CVB * VN * sizeof(float);
    const bool small = small_tensors<T>(d);
    int limit = small ? resident_blocks(dwconv_fwd_stats_kernel<T, K, S, uint32_t>, lds)
                      : resident_blocks(dwconv_fwd_stats_kernel<T, K, S, uint64_t>, lds);
    if (limit > kStatBlocks) limit = kStatBlocks;
    const int cap = limit / by > 0 ? limit / by : 1;
    const int bx = (int)(want < cap ? want : cap);
    if (small)
        hipLaunchKernelGGL((dwconv_fwd_stats_kernel<T, K, S, uint32_t>), dim3(bx, by), dim3(256), lds, s, d, (const T*)x, w, (T*)y, ws);
    else
        hipLaunchKernelGGL((dwconv_fwd_stats_kernel<T, K, S, uint64_t>), dim3(bx, by), dim3(256), lds, s, d, (const T*)x, w, (T*)y, ws);
    if (fin)
        hipLaunchKernelGGL(dwconv_stat_reduce_finalize_kernel, dim3((d.C + kRedCols / 2 - 1) / (kRedCols / 2)), dim3(256), 0, s, bx,
                           d.C, ws, sums, *fin);
    else
        hipLaunchKernelGGL(dwconv_stat_reduce_kernel, dim3((2 * d.C + kRedCols - 1) / kRedCols), dim3(256), 0, s, bx, 2 * d.C, ws, sums);
    return status();
}
template <typename T, int K, int S>
int launch_bwd_data(const DwDims& d, const void* dy, const float* w, void* dx, hipStream_t s) {
    if constexpr (S == 1) {           // convolution of dy (Ho x Wo) with the flipped taps -> dx (H x W)
        constexpr int TW = 4;
        DwDims f = d;
        f.H = d.Ho; f.W = d.Wo; f.Ho = d.H; f.Wo = d.W;
        f.pad_t = K - 1 - d.pad_t; f.pad_l = K - 1 - d.pad_l;
        const int64_t n = (int64_t)f.N * f.Ho * ((f.Wo + TW - 1) / TW) * (f.C / Vec<T>::N);
        if (small_tensors<T>(f))
            hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, 1, TW, true, uint32_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                               f, (const T*)dy, w, (const float*)nullptr, (T*)dx);
        else
            hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, 1, TW, true, uint64_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                               f, (const T*)dy, w, (const float*)nullptr, (T*)dx);
        return status();
    } else if constexpr (S == 2) {
        // quads: rows 2a - pt, 2a + 1 - pt for a = pt / 2 .. (H - 1 + pt) / 2
        const int qrows = ((d.H - 1 + d.pad_t) >> 1) - (d.pad_t >> 1) + 1;
        const int qcols = ((d.W - 1 + d.pad_l) >> 1) - (d.pad_l >> 1) + 1;
        const int CV = d.C / Vec<T>::N;
        const int CVB = CV < 256 ? CV : 256;
        const int QL = 256 / CVB;
        const int by = (CV + CVB - 1) / CVB;
        const int64_t want = ((int64_t)d.N * qrows * qcols + QL - 1) / QL;
        int cap = resident_blocks(dwconv_bwd_data_s2_kernel<T, K>, 0) / by;            // one resident round
        if (cap < 1) cap = 1;
        const int bx = (int)(want < cap ? want : cap);
        hipLaunchKernelGGL((dwconv_bwd_data_s2_kernel<T, K>), dim3(bx, by), dim3(256), 0, s, d, qrows, qcols, (const T*)dy, w,
                           (T*)dx);
        return status();
    } else {
        return STP3_EUNSUP;                                            // (the entry point admits strides 1 and 2 only)
    }
}
template <typename T, int K, int S>
int launch_bwd_weight(const DwDims& d, const void* x, const void* dy, float* dw, float* ws, hipStream_t s, bool param_layout) {
    constexpr int VN = Vec<T>::N;
    const int CV = d.C / VN;
    const int CVB = CV < 256 ? CV : 256;
    const int PL = 256 / CVB;
    const int64_t npix = (int64_t)d.N * d.Ho * ((d.Wo + 3) / 4);       // groups of 4 output pixels
    if (npix >= (1LL << 31)) return STP3_EUNSUP;
    int64_t want = (npix + PL - 1) / PL;
    const int by = (CV + CVB - 1) / CVB;
    const size_t lds = (size_t)PL * K * CVB * VN * sizeof(float);
    const bool small = small_tensors<T>(d);
    int cap = (small ? resident_blocks(dwconv_bwd_weight_kernel<T, K, S, uint32_t>, lds)
                     : resident_blocks(dwconv_bwd_weight_kernel<T, K, S, uint64_t>, lds)) / (by * K);      // (bx * by * K workgroups)
    if (cap > kWgradBlocks) cap = kWgradBlocks;
    if (cap < 1) cap = 1;
    const int bx = (int)(want < cap ? want : cap);
    if (small)
        hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, S, uint32_t>), dim3(bx * by * K), dim3(256), lds, s, d, bx, (const T*)x,
                           (const T*)dy, ws);
    else
        hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, S, uint64_t>), dim3(bx * by * K), dim3(256), lds, s, d, bx, (const T*)x,
                           (const T*)dy, ws);
    const int n = K * K * d.C;
    hipLaunchKernelGGL(dwconv_reduce_partials_kernel, dim3((n + 63) / 64), dim3(256), 0, s, bx, n, ws, dw, param_layout ? d.C : 0);
    return status();
}

#define DISPATCH(FN, ...)                                                                               \
    do {                                                                                                \
        const bool bf = p->dtype == STP3_DTYPE_BF16;                                                    \
        if (p->K == 3 && p->stride == 1) return bf ? FN<uint16_t, 3, 1>(__VA_ARGS__) : FN<float, 3, 1>(__VA_ARGS__); \
        if (p->K == 3 && p->stride == 2) return bf ? FN<uint16_t, 3, 2>(__VA_ARGS__) : FN<float, 3, 2>(__VA_ARGS__); \
        if (p->K == 5 && p->stride == 1) return bf ? FN<uint16_t, 5, 1>(__VA_ARGS__) : FN<float, 5, 1>(__VA_ARGS__); \
        if (p->K == 7) return bf ? FN<uint16_t, 7, 1>(__VA_ARGS__) : FN<float, 7, 1>(__VA_ARGS__);                   \
        return bf ? FN<uint16_t, 5, 2>(__VA_ARGS__) : FN<float, 5, 2>(__VA_ARGS__);                       \
    } while (0)

}  // namespace

extern "C" {

int stp3_dwconv2d_fwd(const stp3_dwconv_dims* p, const void* x, const float* w, void* y, void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!x || !w || !y) return STP3_EINVAL;
    DISPATCH(launch_fwd, d, x, w, (const float*)nullptr, y, (hipStream_t)stream);
}

int stp3_dwconv2d_fwd_bias(const stp3_dwconv_dims* p, const void* x, const float* w, const float* bias, void* y,
                           void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!x || !w || !y) return STP3_EINVAL;
    DISPATCH(launch_fwd, d, x, w, bias, y, (hipStream_t)stream);
}

int stp3_dwconv2d_fwd_stats_workspace(const stp3_dwconv_dims* p, size_t* bytes) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = (size_t)kStatBlocks * 2 * p->C * sizeof(float);
    return STP3_OK;
}

int stp3_dwconv2d_fwd_stats(const stp3_dwconv_dims* p, const void* x, const float* w, void* y, float* sums, void* workspace,
                            size_t workspace_bytes, void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!x || !w || !y || !sums || !workspace) return STP3_EINVAL;
    if (workspace_bytes < (size_t)kStatBlocks * 2 * p->C * sizeof(float)) return STP3_ENOSPACE;
    DISPATCH(launch_fwd_stats, d, x, w, y, sums, (float*)workspace, (hipStream_t)stream);
}

int stp3_dwconv2d_fwd_stats_bn(const stp3_dwconv_dims* p, const void* x, const float* w, void* y, float* sums, double count,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* coef, void* workspace, size_t workspace_bytes, void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!x || !w || !y || !sums || !coef || !workspace || !(count >= 1.0)) return STP3_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return STP3_EINVAL;
    if (workspace_bytes < (size_t)kStatBlocks * 2 * p->C * sizeof(float)) return STP3_ENOSPACE;
    BnFin fin;
    fin.inv_count = (float)(1.0 / count);
    fin.unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
    fin.eps = eps; fin.momentum = momentum; fin.gamma = gamma; fin.beta = beta;
    fin.running_mean = running_mean; fin.running_var = running_var; fin.coef = coef;
    DISPATCH(launch_fwd_stats, d, x, w, y, sums, (float*)workspace, (hipStream_t)stream, &fin);
}

int stp3_dwconv2d_bwd_data(const stp3_dwconv_dims* p, const void* dy, const float* w, void* dx, void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!dy || !w || !dx) return STP3_EINVAL;
    DISPATCH(launch_bwd_data, d, dy, w, dx, (hipStream_t)stream);
}

int stp3_dwconv2d_bwd_weight_workspace(const stp3_dwconv_dims* p, size_t* bytes) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!bytes) return STP3_EINVAL;
    *bytes = (size_t)kWgradBlocks * p->K * p->K * p->C * sizeof(float);
    return STP3_OK;
}

int stp3_dwconv2d_bwd_weight(const stp3_dwconv_dims* p, const void* x, const void* dy, float* dw, void* workspace,
                             size_t workspace_bytes, void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!x || !dy || !dw || !workspace) return STP3_EINVAL;
    if (workspace_bytes < (size_t)kWgradBlocks * p->K * p->K * p->C * sizeof(float)) return STP3_ENOSPACE;
    DISPATCH(launch_bwd_weight, d, x, dy, dw, (float*)workspace, (hipStream_t)stream, false);
}

int stp3_dwconv2d_bwd_weight_oihw(const stp3_dwconv_dims* p, const void* x, const void* dy, float* dw, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    DwDims d; int vec;
    int rc = check(p, &d, &vec);
    if (rc) return rc;
    if (!x || !dy || !dw || !workspace) return STP3_EINVAL;
    if (workspace_bytes < (size_t)kWgradBlocks * p->K * p->K * p->C * sizeof(float)) return STP3_ENOSPACE;
    DISPATCH(launch_bwd_weight, d, x, dy, dw, (float*)workspace, (hipStream_t)stream, true);
}

}  // extern "C"
