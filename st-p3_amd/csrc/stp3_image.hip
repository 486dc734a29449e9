// stp3_image.hip -- camera images from decoded bytes to network input for gfx950.
//
// Replaces, behind stp3_image_prep, the per-image chain of the reference's loader (stp3/datas/NuscenesData.py:236-244):
//     PIL.Image.resize(resize_dims, BILINEAR)  ->  .crop(crop)  ->  ToTensor (/ 255)  ->  Normalize(mean, std)
// 18 times per sample (6 cameras x 3 frames) on 1600 x 900 x 3 bytes each.  BYTE-EXACT with Pillow's resampler: the same
// two passes (horizontal, then vertical) with its fixed-point arithmetic -- coefficients round(w * 2^22) built on the host
// exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do (stp3_amd/datas.py), accumulators starting at 2^21,
// (sum >> 22) clipped to a byte after EACH pass -- so the network sees the numbers it was trained on.
//
// HBM-bound byte work: 4.3 MB in, 0.43-1.3 MB out per image.  One workgroup owns kRows output rows of one image: the input
// rows its vertical taps need are streamed through LDS (16-byte loads of whole rows), reduced horizontally to bytes -- only
// the columns inside the crop -- into an LDS strip, and the vertical pass + normalisation run from the strip; the output
// is written planar (CHW), coalesced, as float32 or bf16.  Pixels of the crop window that lie outside the resized image
// are PIL's zero padding (crop beyond the border), normalised like any other value.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

constexpr int kPrecision = 22;          // Pillow: PRECISION_BITS = 32 - 8 - 2
constexpr int kRows = 4;                // output rows per workgroup
constexpr int kChunk = 4;               // source rows staged per iteration
constexpr int kPre = 5;                 // 16-byte pieces per thread of the chunk in flight (4 x 4 800 bytes: 1 200 pieces)

struct ImageDims {
    int N, H, W, Wr, Hr, left, top, Wo, Ho, ksize_h, ksize_v, out_bf16, strip_rows;
    float mean[3], inv255_unused, std[3];
};

__device__ __forceinline__ int clip8(int v) { return min(max(v >> kPrecision, 0), 255); }

__global__ __launch_bounds__(256) void image_prep_kernel(ImageDims d, const uint8_t* __restrict__ images,
                                                         const int* __restrict__ kk_h, const int* __restrict__ bounds_h,
                                                         const int* __restrict__ kk_v, const int* __restrict__ bounds_v,
                                                         void* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int row_bytes = d.W * 3;
    const int row_pitch = (row_bytes + 15) & ~15;
    uint8_t* rowbuf = smem;                                // kChunk input rows
    uint8_t* strip = smem + kChunk * row_pitch;            // [strip_rows][Wo * 3] horizontally resampled bytes
    // the horizontal coefficients of the window's columns, [Wo][ksize_h] + (first source column, count): read once per
    // workgroup -- fetched per tap from global memory they made the kernel bound by its vector-memory issue rate
    const int kstride = d.ksize_h <= 9 ? 12 : d.ksize_h;    // <= 9 taps: rows padded to 48 bytes for 16-byte reads
    int* ck = reinterpret_cast<int*>(strip + ((d.strip_rows * d.Wo * 3 + 15) & ~15));
    int* cb = ck + d.Wo * kstride;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * kRows, n = blockIdx.y;
    const int rows = min(kRows, d.Ho - r0);
    // input rows the vertical taps of this workgroup's output rows read
    int y0 = d.H, y1 = 0;
    for (int r = 0; r < rows; ++r) {
        const int yr = d.top + r0 + r;
        if (yr < 0 || yr >= d.Hr) continue;
        const int lo = bounds_v[2 * yr], cnt = bounds_v[2 * yr + 1];
        y0 = min(y0, lo);
        y1 = max(y1, lo + cnt);
    }
    const uint8_t* img = images + (size_t)n * d.H * row_bytes;
    const int strip_pitch = d.Wo * 3;
    const bool vec = (row_bytes & 15) == 0 && ((uintptr_t)img & 15) == 0;
    // kChunk source rows at a time: the 16-byte pieces of the NEXT chunk are in flight (registers) while the horizontal
    // pass of the current one runs from LDS -- one row per iteration was latency-bound (0.08 of the HBM rate)
    const int pieces = row_pitch / 16;                     // per row
    uint4 pre[kPre];
    auto fetch = [&](int yc) {                             // pieces tid, tid + 256, ... of the chunk starting at row yc
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const int i = tid + 256 * j;
            const int rr = i / pieces, pp = i - rr * pieces;
            pre[j] = make_uint4(0u, 0u, 0u, 0u);
            if (rr < kChunk && yc + rr < y1) pre[j] = reinterpret_cast<const uint4*>(img + (size_t)(yc + rr) * row_bytes)[pp];
        }
    };
    const bool staged = vec && kChunk * pieces <= 256 * kPre;
    if (staged && y0 < y1) fetch(y0);
    for (int i = tid; i < d.Wo; i += 256) {
        const int xr = d.left + i;
        const bool in = xr >= 0 && xr < d.Wr;
        cb[2 * i] = in ? bounds_h[2 * xr] : 0;
        cb[2 * i + 1] = in ? bounds_h[2 * xr + 1] : 0;     // outside the resized image: no taps, the byte stays 0
        for (int t = 0; t < kstride; ++t) ck[i * kstride + t] = (in && t < d.ksize_h) ? kk_h[xr * d.ksize_h + t] : 0;
    }
    for (int yc = y0; yc < y1; yc += kChunk) {
        const int nrows = min(kChunk, y1 - yc);
        if (staged) {
#pragma unroll
            for (int j = 0; j < kPre; ++j) {
                const int i = tid + 256 * j;
                if (i < kChunk * pieces) reinterpret_cast<uint4*>(rowbuf)[i] = pre[j];
            }
            if (yc + kChunk < y1) fetch(yc + kChunk);
        } else {
            for (int rr = 0; rr < nrows; ++rr)
                for (int i = tid; i < row_bytes; i += 256) rowbuf[rr * row_pitch + i] = img[(size_t)(yc + rr) * row_bytes + i];
        }
        __syncthreads();
        for (int i = tid; i < nrows * d.Wo; i += 256) {     // one thread per (row, column): the three channels share the taps
            const int rr = i / d.Wo, x = i - rr * d.Wo;
            const int lo = cb[2 * x], cnt = cb[2 * x + 1];
            const int* k = ck + x * kstride;
            int s0 = 1 << (kPrecision - 1), s1 = s0, s2 = s0;
            if (kstride == 12) {
                // up to nine taps = 27 consecutive bytes: EIGHT dword reads, realigned to the first byte with funnel shifts,
                // instead of 27 byte reads (the kernel was bound by its LDS instruction count); the coefficients as three
                // 16-byte reads.  Taps beyond `cnt` have coefficient 0; their bytes lie inside the staging buffer.
                const int addr = rr * row_pitch + lo * 3;
                const int sh = (addr & 3) * 8;
                const uint32_t* wp = reinterpret_cast<const uint32_t*>(rowbuf + (addr & ~3));
                uint32_t w[8], v[7];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = wp[j];
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = (uint32_t)(((((uint64_t)w[j + 1]) << 32) | w[j]) >> sh);
                const uint4 ka = reinterpret_cast<const uint4*>(k)[0], kb = reinterpret_cast<const uint4*>(k)[1];
                const int kc = k[8];
                const int kw[9] = {(int)ka.x, (int)ka.y, (int)ka.z, (int)ka.w, (int)kb.x, (int)kb.y, (int)kb.z, (int)kb.w, kc};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int b0 = 3 * t, b1 = 3 * t + 1, b2 = 3 * t + 2;
                    s0 += (int)((v[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu) * kw[t];
                    s1 += (int)((v[b1 >> 2] >> (8 * (b1 & 3))) & 0xffu) * kw[t];
                    s2 += (int)((v[b2 >> 2] >> (8 * (b2 & 3))) & 0xffu) * kw[t];
                }
            } else {
                const uint8_t* src = rowbuf + rr * row_pitch + lo * 3;
                for (int t = 0; t < cnt; ++t) {
                    const int w = k[t];
                    s0 += (int)src[3 * t] * w;
                    s1 += (int)src[3 * t + 1] * w;
                    s2 += (int)src[3 * t + 2] * w;
                }
            }
            uint8_t* dst = strip + (yc - y0 + rr) * strip_pitch + 3 * x;
            dst[0] = cnt ? (uint8_t)clip8(s0) : 0;
            dst[1] = cnt ? (uint8_t)clip8(s1) : 0;
            dst[2] = cnt ? (uint8_t)clip8(s2) : 0;
        }
        __syncthreads();
    }
    // vertical pass + ToTensor + Normalize; planar output, x fastest
    const int per_plane = rows * d.Wo;
    for (int i = tid; i < 3 * per_plane; i += 256) {
        const int c = i / per_plane, rem = i - c * per_plane;
        const int r = rem / d.Wo, x = rem - r * d.Wo;
        const int yr = d.top + r0 + r, xr = d.left + x;
        int v = 0;
        if (yr >= 0 && yr < d.Hr && xr >= 0 && xr < d.Wr) {
            const int lo = bounds_v[2 * yr], cnt = bounds_v[2 * yr + 1];
            const int* k = kk_v + yr * d.ksize_v;
            int ss = 1 << (kPrecision - 1);
            for (int t = 0; t < cnt; ++t) ss += (int)strip[(lo + t - y0) * strip_pitch + x * 3 + c] * k[t];
            v = clip8(ss);
        }
        const float f = ((float)v / 255.0f - d.mean[c]) / d.std[c];
        const size_t o = (((size_t)n * 3 + c) * d.Ho + (r0 + r)) * d.Wo + x;
        if (d.out_bf16) {
            reinterpret_cast<uint16_t*>(out)[o] = (uint16_t)(pack_bf16(f, 0.f) & 0xffffu);
        } else {
            reinterpret_cast<float*>(out)[o] = f;
        }
    }
}

}  // namespace

extern "C" {

int stp3_image_prep_lds_bytes(const stp3_image_dims* p, int32_t strip_rows, size_t* bytes) {
    if (!p || !bytes || strip_rows < 0) return STP3_EINVAL;
    *bytes = (size_t)4 * ((p->W * 3 + 15) & ~15) + (((size_t)strip_rows * p->Wo * 3 + 15) & ~(size_t)15) +     // 4 = kChunk
             (size_t)p->Wo * ((p->ksize_h <= 9 ? 12 : p->ksize_h) + 2) * sizeof(int32_t);
    return STP3_OK;
}

int stp3_image_prep(const stp3_image_dims* p, const uint8_t* images, const int32_t* kk_h, const int32_t* bounds_h,
                    const int32_t* kk_v, const int32_t* bounds_v, int32_t strip_rows, void* out, void* stream) {
    if (!p || !images || !kk_h || !bounds_h || !kk_v || !bounds_v || !out) return STP3_EINVAL;
    if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->Wr <= 0 || p->Hr <= 0 || p->Wo <= 0 || p->Ho <= 0 || p->ksize_h <= 0 ||
        p->ksize_v <= 0 || strip_rows <= 0)
        return STP3_EINVAL;
    if (p->out_dtype != STP3_DTYPE_F32 && p->out_dtype != STP3_DTYPE_BF16) return STP3_EUNSUP;
    if (p->N > 65535 || (int64_t)p->N * p->H * p->W * 3 >= (1LL << 40)) return STP3_EUNSUP;
    size_t lds = 0;
    stp3_image_prep_lds_bytes(p, strip_rows, &lds);
    if (lds > 160 * 1024) return STP3_EUNSUP;              // (a 1600 x 900 source at scale 0.3: 78 KB)
    ImageDims d;
    d.N = p->N; d.H = p->H; d.W = p->W; d.Wr = p->Wr; d.Hr = p->Hr; d.left = p->left; d.top = p->top; d.Wo = p->Wo;
    d.Ho = p->Ho; d.ksize_h = p->ksize_h; d.ksize_v = p->ksize_v; d.out_bf16 = p->out_dtype == STP3_DTYPE_BF16;
    d.strip_rows = strip_rows; d.inv255_unused = 0.f;
    for (int c = 0; c < 3; ++c) { d.mean[c] = p->mean[c]; d.std[c] = p->std[c]; }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&image_prep_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(image_prep_kernel, dim3((p->Ho + kRows - 1) / kRows, p->N), dim3(256), lds, (hipStream_t)stream, d,
                       images, kk_h, bounds_h, kk_v, bounds_v, out);
    e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

int stp3_image_prep_rows_per_workgroup(void) { return kRows; }

}  // extern "C"
