// stp3_se_mlp.hip -- the two tiny fully-connected layers of a squeeze-and-excitation block as single launches.
//
// An MBConv block's gate is  sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2)  on (N, C) / (N, S) tensors with
// N = B*T*cameras = 72, C <= 960, S <= 40 in the EfficientNet-B4 trunk (stp3/models/encoder.py:57-97 drives
// efficientnet_pytorch's MBConvBlock).  Written with torch operators that is ~10 launches forward and ~15 backward
// per block, 22 blocks per step, every one of them launch-latency bound.  Here: one launch forward (a workgroup per
// sample), two backward (per-sample chain, then the weight gradients reduced over the samples in a fixed order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stp3_cdna.h"
#include "stp3_hip.h"

namespace {

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STP3_OK : -(int)e;
}

__device__ __forceinline__ float wave_sum_xor(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// Latency is what these kernels cost (the arithmetic is a few MFLOP): round 3's forward took 29 us per call -- a wave
// reduced kRows = 5 rows of W1 at a time, two rounds of 15 dependent load batches each -- and the per-sample backward 23 us
// with an uncoalesced walk down the columns of W2.  Now every thread owns <= kOwn channels and accumulates ITS part of all
// S dot products at once: S * kOwn independent, coalesced loads in flight per thread, then one pass of wave reductions.
constexpr int kOwn = 4;          // channels per thread and chunk (256 threads: chunks of 1024 channels)
constexpr int kMaxS = 64;        // squeezed channels the register accumulators cover (EfficientNet-B4: <= 40)

// part[s] (this thread's partial sums, s < SR <= kMaxS) -> total[s] in LDS for s < S; `scratch` [4][kMaxS] floats.
// SR (S rounded up to a multiple of 8) is a compile-time bound everywhere: a run-time `s < S` inside the unrolled loops put
// every row's loads behind their own branch -- 40 dependent round trips, 38 us (profiles/r04i: slower than round 3's 29).
// The 16-lane row sums are DPP rotations (one VALU instruction per step, stp3_cdna.h); the 16 rows of the workgroup meet
// in LDS.  (Six __shfl_xor steps per value -- LDS round trips -- made this reduction the longest part of the kernel.)
template <int SR>
__device__ __forceinline__ void block_sums(const float* part, int S, float* scratch, float* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = wave * 4 + (lane >> 4);                 // 0 .. 15
#pragma unroll
    for (int s = 0; s < SR; ++s) {
        const float v = row16_sum(part[s]);
        if ((lane & 15) == 0) scratch[row * kMaxS + s] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < S) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += scratch[r * kMaxS + threadIdx.x];
        total[threadIdx.x] = t;
    }
    __syncthreads();
}

// forward: grid = N, dynamic LDS = S floats
template <int SR>
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(stp3_se_mlp_dims d, const float* __restrict__ pooled_sum,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ z1, float* __restrict__ gate) {
    __shared__ float scratch[16 * kMaxS];
    extern __shared__ float h[];    // [S]: W1 p, then swish(z1)
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int sb = 0; sb < d.S; sb += SR) {                // (one round for S <= 64: every EfficientNet-B4 block)
        const int sn = min(SR, d.S - sb);
        float acc[SR];
#pragma unroll
        for (int s = 0; s < SR; ++s) acc[s] = 0.f;
        for (int c0 = 0; c0 < d.C; c0 += 256 * kOwn) {
            float pc[kOwn];
            int cc[kOwn];
#pragma unroll
            for (int k = 0; k < kOwn; ++k) {
                const int c = c0 + k * 256 + tid;
                cc[k] = min(c, d.C - 1);                       // beyond C: a valid address, times zero
                pc[k] = c < d.C ? pooled_sum[(size_t)n * d.C + c] * d.inv_rows : 0.f;
            }
#pragma unroll
            for (int s = 0; s < SR; ++s) {
                // beyond S: a valid row, result unused.  32-bit element offsets (scalar row offset + the lane's channel):
                // one scalar register per row instead of a 64-bit row pointer (40 of those spilled scalar registers)
                const unsigned roff = (unsigned)(min(sb + s, d.S - 1) * d.C);
#pragma unroll
                for (int k = 0; k < kOwn; ++k) acc[s] = fmaf(w1[roff + (unsigned)cc[k]], pc[k], acc[s]);
            }
        }
        block_sums<SR>(acc, sn, scratch, h + sb);
    }
    for (int s = tid; s < d.S; s += 256) {
        const float z = h[s] + b1[s];
        z1[(size_t)n * d.S + s] = z;
        h[s] = z * sigmoidf_(z);
    }
    __syncthreads();
    // a thread owns channel c and walks ITS row of W2 (S consecutive floats): 16-byte loads when the row length allows
    // (S = 8, 12, 28, 40 of EfficientNet-B4; 6 and 14 take 8-byte loads)
    for (int c = tid; c < d.C; c += 256) {
        float a = b2[c];
        const float* row = w2 + (size_t)c * d.S;
        if ((d.S & 3) == 0 && ((uintptr_t)w2 & 15) == 0) {
            for (int s = 0; s < d.S; s += 4) {
                const float4 w = *reinterpret_cast<const float4*>(row + s);
                a = fmaf(w.x, h[s], a); a = fmaf(w.y, h[s + 1], a); a = fmaf(w.z, h[s + 2], a); a = fmaf(w.w, h[s + 3], a);
            }
        } else if ((d.S & 1) == 0 && ((uintptr_t)w2 & 7) == 0) {
            for (int s = 0; s < d.S; s += 2) {
                const float2 w = *reinterpret_cast<const float2*>(row + s);
                a = fmaf(w.x, h[s], a); a = fmaf(w.y, h[s + 1], a);
            }
        } else {
            for (int s = 0; s < d.S; ++s) a = fmaf(row[s], h[s], a);
        }
        gate[(size_t)n * d.C + c] = sigmoidf_(a);
    }
}

// backward, per sample: dz2 = dgate * gate * (1 - gate); dh = dz2 W2; dz1 = dh * swish'(z1); dpooled = dz1 W1 / rows
// grid = N, dynamic LDS = S floats
template <int SR>
__global__ __launch_bounds__(256) void se_mlp_bwd_sample_kernel(stp3_se_mlp_dims d, const float* __restrict__ dgate,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ z1,
                                                                const float* __restrict__ w1,
                                                                const float* __restrict__ w2, float* __restrict__ dz2,
                                                                float* __restrict__ dz1, float* __restrict__ dpooled) {
    __shared__ float scratch[16 * kMaxS];
    extern __shared__ float g1[];   // [S]: dz2 W2, then dz1 of this sample
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int sb = 0; sb < d.S; sb += SR) {
        const int sn = min(SR, d.S - sb);
        float acc[SR];
#pragma unroll
        for (int s = 0; s < SR; ++s) acc[s] = 0.f;
        // a thread owns channel c: its dz2 times ITS row of W2 (S contiguous floats) is its part of dh
        for (int c = tid; c < d.C; c += 256) {
            const float g = gate[(size_t)n * d.C + c];
            const float v = dgate[(size_t)n * d.C + c] * g * (1.0f - g);
            if (sb == 0) dz2[(size_t)n * d.C + c] = v;
            const unsigned coff = (unsigned)(c * d.S);
            // (c >> 30 is zero; it keeps the clamp a per-lane value: as a uniform one it took a scalar register per row and
            // the 40-row instantiations spilled scalar registers)
            const unsigned smax = (unsigned)(d.S - 1) + (unsigned)(c >> 30);
#pragma unroll
            for (int s = 0; s < SR; ++s) acc[s] = fmaf(v, w2[coff + min((unsigned)(sb + s), smax)], acc[s]);    // beyond S: unused
        }
        block_sums<SR>(acc, sn, scratch, g1 + sb);
    }
    for (int s = tid; s < d.S; s += 256) {
        const float z = z1[(size_t)n * d.S + s];
        const float sg = sigmoidf_(z);
        const float v = g1[s] * sg * (1.0f + z * (1.0f - sg));
        g1[s] = v;
        dz1[(size_t)n * d.S + s] = v;
    }
    __syncthreads();
    for (int c = tid; c < d.C; c += 256) {
        float a = 0.f;
#pragma unroll 8
        for (int s = 0; s < d.S; ++s) a = fmaf(g1[s], w1[(size_t)s * d.C + c], a);
        dpooled[(size_t)n * d.C + c] = a * d.inv_rows;
    }
}

// ---- the same two kernels with their loads issued in explicit batches (S a multiple of 4, S <= SR <= kMaxS) --------------------
// What the kernels above cost is dependent round trips to L2 / the memory-side cache, and how many loads the compiler keeps in
// flight decides their number: the 32-row instantiation of the forward kernel came out fully batched (220 VGPRs, 10 us), the
// 40-row one rolled up (70 VGPRs, 28 us -- profiles/r06s_step_trace.txt), and the loops over a thread's channels (W2 rows in the
// forward pass, W1 columns in the backward pass) were sequential: 4 x 5 round trips of 8 loads.  Here every phase loads a fixed
// batch for ALL of the thread's channels, then multiplies: rows beyond S repeat a valid address and meet a zero (an activation
// vector zero-padded in LDS) or an accumulator nobody reads.  Same sums in the same order as above: bit-equal results.
template <int SR>
__global__ __launch_bounds__(256) void se_mlp_fwd_batched_kernel(stp3_se_mlp_dims d, const float* __restrict__ pooled_sum,
                                                                 const float* __restrict__ w1, const float* __restrict__ b1,
                                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                                 float* __restrict__ z1, float* __restrict__ gate) {
    __shared__ float scratch[16 * kMaxS];
    __shared__ float h[kMaxS];      // W1 p, then swish(z1), zero beyond S
    const int n = blockIdx.x, tid = threadIdx.x;
    float acc[SR];
#pragma unroll
    for (int s = 0; s < SR; ++s) acc[s] = 0.f;
    for (int c0 = 0; c0 < d.C; c0 += 256 * kOwn) {
        float pc[kOwn];
        unsigned cc[kOwn];
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int c = c0 + k * 256 + tid;
            cc[k] = (unsigned)min(c, d.C - 1);                 // beyond C: a valid address, times zero
            pc[k] = c < d.C ? pooled_sum[(size_t)n * d.C + c] * d.inv_rows : 0.f;
        }
#pragma unroll
        for (int s0 = 0; s0 < SR; s0 += 8) {
            float wv[8][kOwn];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned roff = (unsigned)(min(s0 + j, d.S - 1) * d.C);
#pragma unroll
                for (int k = 0; k < kOwn; ++k) wv[j][k] = w1[roff + cc[k]];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < kOwn; ++k) acc[s0 + j] = fmaf(wv[j][k], pc[k], acc[s0 + j]);
        }
    }
    block_sums<SR>(acc, d.S, scratch, h);
    if (tid < kMaxS) {
        float v = 0.f;
        if (tid < d.S) {
            const float z = h[tid] + b1[tid];
            z1[(size_t)n * d.S + tid] = z;
            v = z * sigmoidf_(z);
        }
        h[tid] = v;
    }
    __syncthreads();
    // a thread owns <= kOwn channels per chunk and walks THEIR rows of W2 (S consecutive floats) together, 16 bytes at a time
    for (int c0 = 0; c0 < d.C; c0 += 256 * kOwn) {
        float a[kOwn];
        unsigned roff[kOwn];
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int c = min(c0 + k * 256 + tid, d.C - 1);
            a[k] = b2[c];
            roff[k] = (unsigned)(c * d.S);
        }
#pragma unroll
        for (int s0 = 0; s0 < SR; s0 += 8) {
            float4 wv[2][kOwn];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned so = (unsigned)min(s0 + 4 * q, d.S - 4);
#pragma unroll
                for (int k = 0; k < kOwn; ++k) wv[q][k] = *reinterpret_cast<const float4*>(w2 + roff[k] + so);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float h0 = h[s0 + 4 * q], h1 = h[s0 + 4 * q + 1], h2 = h[s0 + 4 * q + 2], h3 = h[s0 + 4 * q + 3];
#pragma unroll
                for (int k = 0; k < kOwn; ++k) {
                    a[k] = fmaf(wv[q][k].x, h0, a[k]); a[k] = fmaf(wv[q][k].y, h1, a[k]);
                    a[k] = fmaf(wv[q][k].z, h2, a[k]); a[k] = fmaf(wv[q][k].w, h3, a[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int c = c0 + k * 256 + tid;
            if (c < d.C) gate[(size_t)n * d.C + c] = sigmoidf_(a[k]);
        }
    }
}

template <int SR>
__global__ __launch_bounds__(256) void se_mlp_bwd_sample_batched_kernel(stp3_se_mlp_dims d, const float* __restrict__ dgate,
                                                                        const float* __restrict__ gate,
                                                                        const float* __restrict__ z1,
                                                                        const float* __restrict__ w1,
                                                                        const float* __restrict__ w2, float* __restrict__ dz2,
                                                                        float* __restrict__ dz1, float* __restrict__ dpooled) {
    __shared__ float scratch[16 * kMaxS];
    __shared__ float g1[kMaxS];     // dz2 W2, then dz1 of this sample, zero beyond S
    const int n = blockIdx.x, tid = threadIdx.x;
    float acc[SR];
#pragma unroll
    for (int s = 0; s < SR; ++s) acc[s] = 0.f;
    // a thread owns channel c: its dz2 times ITS row of W2 (S contiguous floats) is its part of dh; two channels' rows per batch
    for (int c0 = 0; c0 < d.C; c0 += 256 * 2) {
        float v[2];
        unsigned roff[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = c0 + k * 256 + tid;
            const int cc = min(c, d.C - 1);
            const float g = gate[(size_t)n * d.C + cc];
            v[k] = c < d.C ? dgate[(size_t)n * d.C + cc] * g * (1.0f - g) : 0.f;
            if (c < d.C) dz2[(size_t)n * d.C + c] = v[k];
            roff[k] = (unsigned)(cc * d.S);
        }
        float4 wv[SR / 4][2];
#pragma unroll
        for (int q = 0; q < SR / 4; ++q) {
            const unsigned so = (unsigned)min(4 * q, d.S - 4);
#pragma unroll
            for (int k = 0; k < 2; ++k) wv[q][k] = *reinterpret_cast<const float4*>(w2 + roff[k] + so);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < SR / 4; ++q) {                         // (rows beyond S: accumulators nobody reads)
                acc[4 * q] = fmaf(v[k], wv[q][k].x, acc[4 * q]);         acc[4 * q + 1] = fmaf(v[k], wv[q][k].y, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(v[k], wv[q][k].z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v[k], wv[q][k].w, acc[4 * q + 3]);
            }
    }
    block_sums<SR>(acc, d.S, scratch, g1);
    if (tid < kMaxS) {
        float v = 0.f;
        if (tid < d.S) {
            const float z = z1[(size_t)n * d.S + tid];
            const float sg = sigmoidf_(z);
            v = g1[tid] * sg * (1.0f + z * (1.0f - sg));
            dz1[(size_t)n * d.S + tid] = v;
        }
        g1[tid] = v;
    }
    __syncthreads();
    for (int c0 = 0; c0 < d.C; c0 += 256 * kOwn) {
        float a[kOwn];
        unsigned cc[kOwn];
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            a[k] = 0.f;
            cc[k] = (unsigned)min(c0 + k * 256 + tid, d.C - 1);
        }
#pragma unroll
        for (int s0 = 0; s0 < SR; s0 += 8) {
            float wv[8][kOwn];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned roff = (unsigned)(min(s0 + j, d.S - 1) * d.C);
#pragma unroll
                for (int k = 0; k < kOwn; ++k) wv[j][k] = w1[roff + cc[k]];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gv = g1[s0 + j];
#pragma unroll
                for (int k = 0; k < kOwn; ++k) a[k] = fmaf(gv, wv[j][k], a[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int c = c0 + k * 256 + tid;
            if (c < d.C) dpooled[(size_t)n * d.C + c] = a[k] * d.inv_rows;
        }
    }
}

// backward, weights: workgroup = (64 channels, 8 squeezed channels) x 8 groups of samples; thread (channel, group) adds
// its group's samples n = group, group + 8, ... in ascending order, the eight groups are added in order: deterministic
//   dw2[c][s] = sum_n dz2[n][c] swish(z1[n][s]);  db2[c] = sum_n dz2[n][c]
//   dw1[s][c] = sum_n dz1[n][s] pooled[n][c];      db1[s] = sum_n dz1[n][s]   (workgroup (0, *))
// LDS = 2 * N * 8 floats (swish(z1) and dz1 of all samples for the workgroup's 8 squeezed channels) + the group sums.
// (Round 3: one thread per channel walked all N samples, two dependent-latency loads per sample: 17 us per call.)
constexpr int kSChunk = 8;
constexpr int kWgtChan = 64;
constexpr int kWgtGroups = 8;
constexpr int kWgtThreads = kWgtChan * kWgtGroups;

__global__ __launch_bounds__(kWgtThreads) void se_mlp_bwd_weight_kernel(stp3_se_mlp_dims d,
                                                                        const float* __restrict__ pooled_sum,
                                                                        const float* __restrict__ z1,
                                                                        const float* __restrict__ dz2,
                                                                        const float* __restrict__ dz1,
                                                                        float* __restrict__ dw1, float* __restrict__ db1,
                                                                        float* __restrict__ dw2, float* __restrict__ db2) {
    extern __shared__ float smem[];
    float* hs = smem;                                   // [N][8]
    float* g1 = smem + (size_t)d.N * kSChunk;           // [N][8]
    float* red = g1 + (size_t)d.N * kSChunk;            // [groups][2 * 8 + 1][64]
    const int tid = threadIdx.x, cl = tid % kWgtChan, grp = tid / kWgtChan;
    const int s0 = blockIdx.y * kSChunk;
    for (int i = tid; i < d.N * kSChunk; i += kWgtThreads) {
        const int n = i / kSChunk, s = s0 + i % kSChunk;
        const float z = s < d.S ? z1[n * d.S + s] : 0.f;
        hs[i] = z * sigmoidf_(z);
        g1[i] = s < d.S ? dz1[n * d.S + s] : 0.f;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < kSChunk && s0 + tid < d.S) {
        float a = 0.f;
        for (int n = 0; n < d.N; ++n) a += g1[n * kSChunk + tid];
        db1[s0 + tid] = a;
    }
    const int c = blockIdx.x * kWgtChan + cl;
    float sb = 0.f;
    float a2[kSChunk], a1[kSChunk];
#pragma unroll
    for (int k = 0; k < kSChunk; ++k) a2[k] = a1[k] = 0.f;
    if (c < d.C) {
#pragma unroll 6
        for (int n = grp; n < d.N; n += kWgtGroups) {
            const float x2 = dz2[(size_t)n * d.C + c];
            const float pc = pooled_sum[(size_t)n * d.C + c] * d.inv_rows;
            sb += x2;
#pragma unroll
            for (int k = 0; k < kSChunk; ++k) {
                a2[k] = fmaf(x2, hs[n * kSChunk + k], a2[k]);
                a1[k] = fmaf(g1[n * kSChunk + k], pc, a1[k]);
            }
        }
    }
    constexpr int kVals = 2 * kSChunk + 1;
#pragma unroll
    for (int k = 0; k < kSChunk; ++k) {
        red[(grp * kVals + k) * kWgtChan + cl] = a2[k];
        red[(grp * kVals + kSChunk + k) * kWgtChan + cl] = a1[k];
    }
    red[(grp * kVals + 2 * kSChunk) * kWgtChan + cl] = sb;
    __syncthreads();
    if (grp != 0 || c >= d.C) return;
#pragma unroll
    for (int k = 0; k < kVals; ++k) {
        float t = red[k * kWgtChan + cl];
#pragma unroll
        for (int g = 1; g < kWgtGroups; ++g) t += red[(g * kVals + k) * kWgtChan + cl];
        if (k < kSChunk) {
            if (s0 + k < d.S) dw2[(size_t)c * d.S + s0 + k] = t;
        } else if (k < 2 * kSChunk) {
            if (s0 + k - kSChunk < d.S) dw1[(size_t)(s0 + k - kSChunk) * d.C + c] = t;
        } else if (blockIdx.y == 0) {
            db2[c] = t;
        }
    }
}

inline int check(const stp3_se_mlp_dims* d) {
    if (!d) return STP3_EINVAL;
    if (d->N <= 0 || d->C <= 0 || d->S <= 0) return STP3_EINVAL;
    // (the weight-gradient kernel's LDS: the samples' 8-channel slices + the group sums, within the 64 KB a launch may ask for)
    if ((size_t)d->S * 4 > 32 * 1024 ||
        ((size_t)2 * d->N * kSChunk + (size_t)kWgtGroups * (2 * kSChunk + 1) * kWgtChan) * 4 > 64 * 1024)
        return STP3_EUNSUP;
    if ((int64_t)d->S * d->C >= (1LL << 31)) return STP3_EUNSUP;       // 32-bit element offsets into the weight matrices
    return STP3_OK;
}

// the batched kernels: whole 16-byte pieces of the W2 rows (S a multiple of 4, the matrix 16-byte aligned), S within the
// register accumulators
inline bool batched(const stp3_se_mlp_dims* d, const float* w2) {
    return d->S % 4 == 0 && d->S <= kMaxS && ((uintptr_t)w2 & 15) == 0;
}

}  // namespace

extern "C" {

int stp3_se_mlp_fwd(const stp3_se_mlp_dims* dims, const float* pooled_sum, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* z1, float* gate, void* stream) {
    int rc = check(dims);
    if (rc) return rc;
    if (!pooled_sum || !w1 || !b1 || !w2 || !b2 || !z1 || !gate) return STP3_EINVAL;
    const size_t lds = (size_t)dims->S * 4;
    hipStream_t s = (hipStream_t)stream;
    if (batched(dims, w2)) {
#define STP3_SE_FWDB(SR) hipLaunchKernelGGL(se_mlp_fwd_batched_kernel<SR>, dim3(dims->N), dim3(256), 0, s, *dims, pooled_sum, w1, b1, w2, b2, z1, gate)
        switch ((dims->S + 7) / 8) {
            case 1: STP3_SE_FWDB(8); break;
            case 2: STP3_SE_FWDB(16); break;
            case 3: STP3_SE_FWDB(24); break;
            case 4: STP3_SE_FWDB(32); break;
            case 5: STP3_SE_FWDB(40); break;
            case 6: STP3_SE_FWDB(48); break;
            default: STP3_SE_FWDB(64); break;
        }
#undef STP3_SE_FWDB
        return launch_status();
    }
#define STP3_SE_FWD(SR) hipLaunchKernelGGL(se_mlp_fwd_kernel<SR>, dim3(dims->N), dim3(256), lds, s, *dims, pooled_sum, w1, b1, w2, b2, z1, gate)
    switch ((dims->S + 7) / 8) {
        case 1: STP3_SE_FWD(8); break;
        case 2: STP3_SE_FWD(16); break;
        case 3: STP3_SE_FWD(24); break;
        case 4: STP3_SE_FWD(32); break;
        case 5: STP3_SE_FWD(40); break;
        case 6: STP3_SE_FWD(48); break;
        default: STP3_SE_FWD(64); break;
    }
#undef STP3_SE_FWD
    return launch_status();
}

int stp3_se_mlp_bwd(const stp3_se_mlp_dims* dims, const float* dgate, const float* gate, const float* pooled_sum,
                    const float* z1, const float* w1, const float* w2, float* dz2, float* dz1, float* dpooled,
                    float* dw1, float* db1, float* dw2, float* db2, void* stream) {
    int rc = check(dims);
    if (rc) return rc;
    if (!dgate || !gate || !pooled_sum || !z1 || !w1 || !w2 || !dz2 || !dz1 || !dpooled || !dw1 || !db1 || !dw2 || !db2)
        return STP3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)dims->S * 4;
    if (batched(dims, w2)) {
#define STP3_SE_BWDB(SR) hipLaunchKernelGGL(se_mlp_bwd_sample_batched_kernel<SR>, dim3(dims->N), dim3(256), 0, s, *dims, dgate, gate, z1, w1, w2, dz2, dz1, dpooled)
        switch ((dims->S + 7) / 8) {
            case 1: STP3_SE_BWDB(8); break;
            case 2: STP3_SE_BWDB(16); break;
            case 3: STP3_SE_BWDB(24); break;
            case 4: STP3_SE_BWDB(32); break;
            case 5: STP3_SE_BWDB(40); break;
            case 6: STP3_SE_BWDB(48); break;
            default: STP3_SE_BWDB(64); break;
        }
#undef STP3_SE_BWDB
    } else {
#define STP3_SE_BWD(SR) hipLaunchKernelGGL(se_mlp_bwd_sample_kernel<SR>, dim3(dims->N), dim3(256), lds, s, *dims, dgate, gate, z1, w1, w2, dz2, dz1, dpooled)
        switch ((dims->S + 7) / 8) {
            case 1: STP3_SE_BWD(8); break;
            case 2: STP3_SE_BWD(16); break;
            case 3: STP3_SE_BWD(24); break;
            case 4: STP3_SE_BWD(32); break;
            case 5: STP3_SE_BWD(40); break;
            case 6: STP3_SE_BWD(48); break;
            default: STP3_SE_BWD(64); break;
        }
#undef STP3_SE_BWD
    }
    hipLaunchKernelGGL(se_mlp_bwd_weight_kernel, dim3((dims->C + kWgtChan - 1) / kWgtChan, (dims->S + kSChunk - 1) / kSChunk),
                       dim3(kWgtThreads),
                       ((size_t)2 * dims->N * kSChunk + (size_t)kWgtGroups * (2 * kSChunk + 1) * kWgtChan) * 4, s, *dims,
                       pooled_sum, z1, dz2, dz1, dw1, db1, dw2, db2);
    return launch_status();
}

}  // extern "C"
