// TEST INFRASTRUCTURE -- execution engine of the CPU stand-in for HIP (see hip/hip_runtime.h in this directory).
//
// A launch hands its workgroups to a small pool of OS threads.  A pool thread runs one workgroup at a time: every HIP
// thread of the group is a fiber (ucontext) with its own stack; the fibers run round-robin until each has returned.
// A fiber that reaches a barrier parks until the barrier's generation moves; if a full pass over the live fibers
// resumes none and none has finished, the workgroup is deadlocked (a barrier or wave intrinsic reached by only part
// of its threads) and the process aborts with a message.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <thread>

namespace hipcpu {

thread_local Ctx* cur = nullptr;
alignas(64) thread_local unsigned char g_dyn_lds[kDynLds];

namespace {

constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    void* stack = nullptr;
    Ctx c;
    bool done = false;
};

struct Worker {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    const std::function<void()>* kernel = nullptr;
    int running = -1;
};
thread_local Worker* wk = nullptr;

void trampoline() {
    Worker* w = wk;
    Fiber& f = w->fibers[w->running];
    (*w->kernel)();
    f.done = true;
    f.c.block->waves[f.c.wave].live &= ~(1ull << f.c.lane);
    f.c.block->waves[f.c.wave].bar.drop();
    f.c.block->bar.drop();
    swapcontext(&f.ctx, &w->sched);
}

void run_block(Worker& w, Block& blk, dim3 grid, dim3 block, long long b, int nthreads, int nwaves, size_t lds_bytes) {
    // LDS is not initialised on the GPU: poison the dynamic part (NaN as float, -1 as int) so that a read of a
    // location the block never wrote cannot go unnoticed
    std::memset(g_dyn_lds, 0xFF, lds_bytes ? lds_bytes : 0);
    blk.bar.reset(nthreads);
    for (int v = 0; v < nwaves; ++v) {
        const int lanes = std::min(kWave, nthreads - v * kWave);
        blk.waves[v].bar.reset(lanes);
        blk.waves[v].live = lanes == 64 ? ~0ull : ((1ull << lanes) - 1);
        std::memset(blk.waves[v].slot, 0, sizeof(blk.waves[v].slot));
    }
    const dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = w.fibers[t];
        f.done = false;
        f.c = Ctx();
        f.c.block = &blk;
        f.c.bdim = block;
        f.c.gdim = grid;
        f.c.bid = bid;
        f.c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.c.lane = t % kWave;
        f.c.wave = t / kWave;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
    }
    // The order in which runnable fibers are resumed is not part of HIP's contract.  HIPCPU_ORDER=reverse / random
    // changes it, so that a missing barrier (an LDS / global hand-off that only works because thread i happened to run
    // before thread j) shows up as a wrong result under one of the orders.
    static const int order_mode = [] {
        const char* e = std::getenv("HIPCPU_ORDER");
        return !e ? 0 : !std::strcmp(e, "reverse") ? 1 : !std::strcmp(e, "random") ? 2 : 0;
    }();
    std::vector<int> order(nthreads);
    for (int t = 0; t < nthreads; ++t) order[t] = order_mode == 1 ? nthreads - 1 - t : t;
    unsigned rng = 12345u + (unsigned)b * 2654435761u;
    int left = nthreads;
    while (left > 0) {
        bool progressed = false;
        if (order_mode == 2)
            for (int i = nthreads - 1; i > 0; --i) {                      // Fisher-Yates with an LCG, new order every pass
                rng = rng * 1664525u + 1013904223u;
                std::swap(order[i], order[(rng >> 8) % (unsigned)(i + 1)]);
            }
        for (int k = 0; k < nthreads; ++k) {
            const int t = order[k];
            Fiber& f = w.fibers[t];
            if (f.done) continue;
            if (f.c.waiting_on && f.c.waiting_on->gen == f.c.wait_gen) continue;     // still parked
            w.running = t;
            cur = &f.c;
            swapcontext(&w.sched, &f.ctx);
            progressed = true;
            if (f.done) --left;
        }
        if (!progressed) die("deadlock inside a workgroup: a barrier or wave intrinsic was reached by only part of its threads");
    }
    cur = nullptr;
}

}  // namespace

void yield_to_scheduler() {
    Worker* w = wk;
    swapcontext(&w->fibers[w->running].ctx, &w->sched);
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& kernel) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) die("block size out of range");
    if (lds_bytes > kDynLds) die("dynamic LDS request above 160 KB");
    const int nwaves = (nthreads + kWave - 1) / kWave;
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    if (nblocks <= 0) return;
    static const int pool = [] {
        const char* e = std::getenv("HIPCPU_THREADS");
        int n = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }();
    const int nworkers = (int)std::min<long long>(pool, nblocks);
    std::atomic<long long> next{0};
    auto body = [&]() {
        Worker w;
        w.kernel = &kernel;
        w.fibers.resize(nthreads);
        for (auto& f : w.fibers) {
            f.stack = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (f.stack == MAP_FAILED) die("cannot map a fiber stack");
        }
        wk = &w;
        Block blk;
        blk.waves = std::vector<Wave>(nwaves);
        for (long long b = next.fetch_add(1); b < nblocks; b = next.fetch_add(1))
            run_block(w, blk, grid, block, b, nthreads, nwaves, lds_bytes);
        wk = nullptr;
        for (auto& f : w.fibers) munmap(f.stack, kStack);
    };
    std::vector<std::thread> threads;
    for (int i = 1; i < nworkers; ++i) threads.emplace_back(body);
    body();
    for (auto& t : threads) t.join();
}

}  // namespace hipcpu
