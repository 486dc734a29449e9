// Measurement aid (not part of the library): what the store PATTERN of a streaming kernel costs on gfx950.
// Writes an [M][Cout] bf16 tensor (Cout = 144: pixel rows of 288 bytes) with persistent waves in five patterns and
// prints us / TB/s for each:
//   0  linear: a wave stores 1 KB contiguous per instruction (the ceiling)
//   1  pieces of 32 B: wave = (pixel group, 32-channel block); lane (pixel, h) stores 2 x 16 B; the lanes l, l + 32 adjacent
//   2  pieces of 64 B: the same blocks, four lanes per pixel adjacent (after a lane exchange)
//   3  pixel rows of 128 B: wave = (pixel group, 64-channel block), 8 lanes per pixel (after an LDS transposition)
//   4  like 3 with whole pixel rows (288 B) per wave: 18 lanes per pixel
//     hipcc --offload-arch=gfx950 -O3 scripts/probe_stores.hip -o scripts/probe_stores && scripts/probe_stores
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kCout = 144, kRow = kCout * 2;

template <int P>
__global__ __launch_bounds__(256) void store_kernel(uint8_t* __restrict__ y, int M, int nwaves) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint4 v = make_uint4(slot, lane, 3u, 4u);
    const int ntiles = M / 32;
    if (P == 0) {
        const size_t total = (size_t)M * kRow / 1024;                  // 1-KB chunks
        for (size_t c = slot; c < total; c += nwaves) *reinterpret_cast<uint4*>(y + c * 1024 + lane * 16) = v;
    } else if (P == 1 || P == 2) {
        const int cblocks = 5, ngroups = nwaves / cblocks;
        const int group = slot / cblocks, cb = slot - group * cblocks;
        if (group >= ngroups) return;
        for (int t = group; t < ntiles; t += ngroups) {
            if (P == 1) {
                const int px = lane & 31, h = lane >> 5;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = cb * 64 + s * 32 + h * 16;
                    if (off < kRow) *reinterpret_cast<uint4*>(y + (size_t)(t * 32 + px) * kRow + off) = v;
                }
            } else {
                // instruction k: pixel 2 i + k, lanes (2 i, h), (2 i + 1, h): piece s = lane parity
                const int i = (lane & 31) >> 1, s = lane & 1, h = lane >> 5;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int off = cb * 64 + s * 32 + h * 16;
                    if (off < kRow) *reinterpret_cast<uint4*>(y + (size_t)(t * 32 + 2 * i + k) * kRow + off) = v;
                }
            }
        }
    } else if (P == 3) {
        const int cblocks = 3, ngroups = nwaves / cblocks;
        const int group = slot / cblocks, cb = slot - group * cblocks;
        if (group >= ngroups) return;
        for (int t = group; t < ntiles; t += ngroups) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = (lane >> 3) + 8 * i, off = cb * 128 + (lane & 7) * 16;
                if (off < kRow) *reinterpret_cast<uint4*>(y + (size_t)(t * 32 + p) * kRow + off) = v;
            }
        }
    } else {
        for (int t = slot; t < ntiles; t += nwaves) {
            // 32 pixels x 18 pieces = 576 pieces = 9 instructions
#pragma unroll
            for (int i = 0; i < 9; ++i) *reinterpret_cast<uint4*>(y + (size_t)t * 32 * kRow + (size_t)(i * 64 + lane) * 16) = v;
        }
    }
}

template <int P>
void run(uint8_t* y, int M, int nwg, const char* what) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(nwg), dim3(256), 0, 0, y, M, nwg * 4);
    hipEventRecord(a, 0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(nwg), dim3(256), 0, 0, y, M, nwg * 4);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters, bytes = (double)M * kRow;
    printf("  pattern %d (%s): %7.1f us  %5.2f TB/s\n", P, what, us, bytes / us / 1e6);
}

int main() {
    for (int M : {72 * 56 * 120, 72 * 112 * 240}) {
        uint8_t* y = nullptr;
        hipMalloc(&y, (size_t)M * kRow);
        hipMemset(y, 0, (size_t)M * kRow);
        for (int nwg : {512, 1024, 2048}) {
            printf("M = %d pixels (%.0f MiB), %d workgroups of 4 waves\n", M, (double)M * kRow / 1048576.0, nwg);
            run<0>(y, M, nwg, "linear 1 KB per wave instruction");
            run<1>(y, M, nwg, "32-byte pieces");
            run<2>(y, M, nwg, "64-byte pieces");
            run<3>(y, M, nwg, "128-byte pixel rows of a 64-channel block");
            run<4>(y, M, nwg, "whole pixel rows");
        }
        hipFree(y);
    }
    return 0;
}
