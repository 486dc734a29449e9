// TEST INFRASTRUCTURE -- execution engine of the CPU stand-in for HIP (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

namespace hipcpu {

thread_local Ctx tls;
alignas(64) unsigned char g_dyn_lds[kDynLds];

// One launch: `nthreads` OS threads walk the grid together, block after block.  An outer (never dropped) barrier
// separates the blocks so that static / dynamic shared memory of one block is not reused while threads of the
// previous block are still running; a thread whose kernel body returned drops out of the block's barriers.
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& kernel) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) die("block size out of range");
    if (lds_bytes > kDynLds) die("dynamic LDS request above 160 KB");
    const int nwaves = (nthreads + kWave - 1) / kWave;
    Block blk;
    blk.waves = std::vector<Wave>(nwaves);
    Barrier outer;
    outer.reset(nthreads);
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    auto worker = [&](int t) {
        Ctx& c = tls;
        c.block = &blk;
        c.bdim = block;
        c.gdim = grid;
        c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        c.lane = t % kWave;
        c.wave = t / kWave;
        for (long long b = 0; b < nblocks; ++b) {
            c.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
            if (t == 0) {
                blk.bar.reset(nthreads);
                for (int w = 0; w < nwaves; ++w) {
                    const int live = std::min(kWave, nthreads - w * kWave);
                    blk.waves[w].bar.reset(live);
                    std::memset(blk.waves[w].slot, 0, sizeof(blk.waves[w].slot));
                }
            }
            outer.wait();
            kernel();
            blk.waves[c.wave].bar.drop();
            blk.bar.drop();
            outer.wait();
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
}

}  // namespace hipcpu
