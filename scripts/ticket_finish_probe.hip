// What does a "last arriving workgroup finishes the reduction" tail cost on the MI355X, next to the separate
// second-stage launch it would replace?  Stand-alone probe (no library code): a streaming producer of B partial rows
// (the shape of bn_stats / bn_bwd_reduce: every workgroup reads its share of a large array and leaves one row of
// `width` floats), finished either by
//   (a) a second kernel that sums the B rows                              (what the library does today),
//   (b) two ticket levels inside the producer: the last workgroup of every group of 32 sums the group's rows, the
//       last group finisher sums the group rows                            (fixed order: deterministic),
//   (c) one ticket level: the last workgroup sums all B rows.
// Each variant is captured `reps` times back to back into a hipGraph (like the training step) and replayed; the
// probe prints microseconds per producer(+finish), checks (b) and (c) against (a) (same sums, another association)
// and every replay against the first one bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ticket_probe scripts/ticket_finish_probe.hip && /tmp/ticket_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

constexpr int kThreads = 256;
constexpr int kGroup = 32;

// every workgroup sums its strided share of src into one row of `width` floats (column c: elements = c mod width)
__device__ void produce_row(const float4* __restrict__ src, size_t nvec, int width, float* __restrict__ row,
                            float* red) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const size_t step = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * step < nvec; i += 4 * step) {                     // four independent 16-byte loads in flight
        const float4 u = src[i], v = src[i + step], w = src[i + 2 * step], x = src[i + 3 * step];
        a0 += (u.x + u.y) + (u.z + u.w);
        a1 += (v.x + v.y) + (v.z + v.w);
        a2 += (w.x + w.y) + (w.z + w.w);
        a3 += (x.x + x.y) + (x.z + x.w);
    }
    for (; i < nvec; i += step) {
        const float4 u = src[i];
        a0 += (u.x + u.y) + (u.z + u.w);
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int c = threadIdx.x; c < width; c += kThreads) row[c] = red[c % kThreads] + (float)c;
}

__global__ __launch_bounds__(kThreads) void produce_kernel(const float4* __restrict__ src, size_t nvec, int width,
                                                           float* __restrict__ partial) {
    __shared__ float red[kThreads];
    produce_row(src, nvec, width, partial + (size_t)blockIdx.x * width, red);
}

// out[c] = sum over rows, the library's second stage (bn_reduce_partials_kernel): 8 columns x 32 row lanes per
// workgroup, four loads in flight, tree over the lanes -- in double
__global__ __launch_bounds__(kThreads) void second_stage_kernel(int parts, int width, const float* __restrict__ partial,
                                                                float* __restrict__ out) {
    __shared__ double red[kThreads];
    const int il = threadIdx.x % 8, pl = threadIdx.x / 8;
    const int c = blockIdx.x * 8 + il;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < width) {
        int p = pl;
        for (; p + 96 < parts; p += 128) {
            const float a = partial[(size_t)p * width + c], b = partial[(size_t)(p + 32) * width + c];
            const float d = partial[(size_t)(p + 64) * width + c], e = partial[(size_t)(p + 96) * width + c];
            s0 += (double)a; s1 += (double)b; s2 += (double)d; s3 += (double)e;
        }
        for (; p < parts; p += 32) s0 += (double)partial[(size_t)p * width + c];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if (pl < st) red[threadIdx.x] += red[threadIdx.x + st * 8];
        __syncthreads();
    }
    if (pl == 0 && c < width) out[c] = (float)red[il];
}

// true in every thread of the workgroup that took the last of `expected` tickets; resets the counter for the next launch
// FENCE_ALL: every thread fences its own stores (the textbook form); otherwise the workgroup barrier orders the
// stores before thread 0, whose single agent-scope fence then covers them (one L2 write-back per workgroup)
template <bool FENCE_ALL>
__device__ bool last_arrival(int* counter, int expected, int* flag) {
    if (FENCE_ALL) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!FENCE_ALL) __threadfence();
        const int t = atomicAdd(counter, 1);
        const int last = t == expected - 1;
        if (last) *counter = 0;
        *flag = last;
    }
    __syncthreads();
    const bool last = *flag != 0;
    if (last) __threadfence();             // ... and the other workgroups' rows before they are read
    return last;
}

template <bool FENCE_ALL>
__global__ __launch_bounds__(kThreads) void produce_ticket2_kernel(const float4* __restrict__ src, size_t nvec, int width,
                                                                   float* __restrict__ partial,
                                                                   double* __restrict__ group_rows, int* counters,
                                                                   float* __restrict__ out) {
    __shared__ float red[kThreads];
    __shared__ int flag;
    const int b = blockIdx.x, nb = gridDim.x;
    produce_row(src, nvec, width, partial + (size_t)b * width, red);
    const int g = b / kGroup, ng = (nb + kGroup - 1) / kGroup;
    const int g0 = g * kGroup, gsize = min(kGroup, nb - g0);
    if (!last_arrival<FENCE_ALL>(counters + 1 + g, gsize, &flag)) return;
    const float* rows = FENCE_ALL ? partial : partial;   // plain loads: the acquire fence invalidated the caches
    for (int c = threadIdx.x; c < width; c += kThreads) {
        double t = 0.0;
        for (int p = g0; p < g0 + gsize; ++p) t += (double)rows[(size_t)p * width + c];
        group_rows[(size_t)g * width + c] = t;
    }
    if (!last_arrival<FENCE_ALL>(counters, ng, &flag)) return;
    const double* grows = group_rows;
    for (int c = threadIdx.x; c < width; c += kThreads) {
        double s = 0.0;
        for (int q = 0; q < ng; ++q) s += grows[(size_t)q * width + c];
        out[c] = (float)s;
    }
}

template <bool FENCE_ALL>
__global__ __launch_bounds__(kThreads) void produce_ticket1_kernel(const float4* __restrict__ src, size_t nvec, int width,
                                                                   float* __restrict__ partial, int* counters,
                                                                   float* __restrict__ out) {
    __shared__ float red[kThreads];
    __shared__ int flag;
    const int b = blockIdx.x, nb = gridDim.x;
    produce_row(src, nvec, width, partial + (size_t)b * width, red);
    if (!last_arrival<FENCE_ALL>(counters, nb, &flag)) return;
    const float* rows = FENCE_ALL ? partial : partial;   // plain loads: the acquire fence invalidated the caches
    for (int c = threadIdx.x; c < width; c += kThreads) {
        double s = 0.0;
        for (int g = 0; g < nb; g += kGroup) {
            double t = 0.0;
            const int e = min(g + kGroup, nb);
            for (int p = g; p < e; ++p) t += (double)rows[(size_t)p * width + c];
            s += t;
        }
        out[c] = (float)s;
    }
}

struct Case { int blocks, width; size_t mbytes; };

int main() {
    const Case cases[] = {{144, 128, 4}, {504, 384, 24}, {504, 1344, 24}, {1024, 192, 64}, {1024, 384, 245}, {1020, 768, 120}};
    const int reps = 50, replays = 20;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    size_t max_bytes = 256u << 20;
    float4* src;
    CHECK(hipMalloc(&src, max_bytes));
    {
        std::vector<float> h(max_bytes / 4);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)((x >> 9) & 0xff) * (1.f / 64.f) - 2.f; }
        CHECK(hipMemcpy(src, h.data(), max_bytes, hipMemcpyHostToDevice));
    }
    float *partial, *out_a, *out_b, *out_c;
    double* group_rows;
    int* counters;
    CHECK(hipMalloc(&partial, (size_t)1100 * 1400 * 4));
    CHECK(hipMalloc(&group_rows, (size_t)64 * 1400 * 8));
    CHECK(hipMalloc(&out_a, 1400 * 4));
    CHECK(hipMalloc(&out_b, 1400 * 4));
    CHECK(hipMalloc(&out_c, 1400 * 4));
    CHECK(hipMalloc(&counters, 4096));
    CHECK(hipMemset(counters, 0, 4096));
    printf("%-26s %10s %10s %10s %10s %10s   (us per producer + finish, %d chained in a hipGraph)\n",
           "blocks x width, MB read", "producer", "2 launches", "2 tickets", "1 ticket", "2t 1fence", reps);
    for (const Case& c : cases) {
        const size_t nvec = (c.mbytes << 20) / 16;
        double us[5] = {0, 0, 0, 0, 0};
        int bad[5] = {0, 0, 0, 0, 0};
        std::vector<float> ref(c.width), got(c.width), first(c.width);
        for (int variant = 0; variant < 5; ++variant) {
            hipGraph_t graph;
            hipGraphExec_t exec;
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int r = 0; r < reps; ++r) {
                if (variant == 0) {
                    hipLaunchKernelGGL(produce_kernel, dim3(c.blocks), dim3(kThreads), 0, s, src, nvec, c.width, partial);
                    hipLaunchKernelGGL(second_stage_kernel, dim3((c.width + 7) / 8), dim3(kThreads), 0, s,
                                       c.blocks, c.width, partial, out_a);
                } else if (variant == 1) {
                    hipLaunchKernelGGL(produce_ticket2_kernel<true>, dim3(c.blocks), dim3(kThreads), 0, s, src, nvec,
                                       c.width, partial, group_rows, counters, out_b);
                } else if (variant == 2) {
                    hipLaunchKernelGGL(produce_ticket1_kernel<true>, dim3(c.blocks), dim3(kThreads), 0, s, src, nvec,
                                       c.width, partial, counters, out_c);
                } else if (variant == 3) {
                    hipLaunchKernelGGL(produce_ticket2_kernel<false>, dim3(c.blocks), dim3(kThreads), 0, s, src, nvec,
                                       c.width, partial, group_rows, counters, out_b);
                } else {
                    hipLaunchKernelGGL(produce_kernel, dim3(c.blocks), dim3(kThreads), 0, s, src, nvec, c.width, partial);
                }
            }
            CHECK(hipStreamEndCapture(s, &graph));
            CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            CHECK(hipGraphLaunch(exec, s));                                   // warm
            CHECK(hipStreamSynchronize(s));
            float best = 1e30f;
            for (int k = 0; k < replays; ++k) {
                float* out = variant == 0 ? out_a : variant == 2 ? out_c : out_b;
                CHECK(hipMemsetAsync(out, 0xff, c.width * 4, s));
                CHECK(hipEventRecord(e0, s));
                CHECK(hipGraphLaunch(exec, s));
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CHECK(hipMemcpy(got.data(), out, c.width * 4, hipMemcpyDeviceToHost));
                if (variant == 4) continue;                                       // producer alone: nothing to check
                if (k == 0) {
                    if (variant == 0) ref = got;
                    first = got;
                    for (int j = 0; j < c.width; ++j)                              // same sums, another association
                        if (!(fabs((double)got[j] - ref[j]) <= 1e-5 * fabs((double)ref[j]) + 1e-3)) { ++bad[variant]; break; }
                }
                if (memcmp(first.data(), got.data(), c.width * 4) != 0) ++bad[variant];   // replay == replay, bit for bit
            }
            us[variant] = best * 1e3 / reps;
            CHECK(hipGraphExecDestroy(exec));
            CHECK(hipGraphDestroy(graph));
        }
        char name[64];
        snprintf(name, sizeof name, "%4d x %4d, %3zu MB", c.blocks, c.width, c.mbytes);
        printf("%-26s %10.2f %10.2f %10.2f %10.2f %10.2f   mismatches: %d %d %d %d\n", name, us[4], us[0], us[1], us[2],
               us[3], bad[0], bad[1], bad[2], bad[3]);
    }
    int left[8];
    CHECK(hipMemcpy(left, counters, sizeof left, hipMemcpyDeviceToHost));
    printf("counters after the run (must be zero): %d %d %d %d\n", left[0], left[1], left[2], left[3]);
    return 0;
}
