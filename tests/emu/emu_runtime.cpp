// tests/emu/emu_runtime.cpp -- fiber scheduler of the wave64 SIMT emulator
// (TEST INFRASTRUCTURE, see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <mutex>
#include <vector>

namespace emu {

struct WaveSync {
    int alive = 0, arrived = 0;
    unsigned long long live_mask = 0;
    uint64_t phase = 0;
    int64_t slot[2][64];
    unsigned long long bal[2] = { 0, 0 };
    void complete() {  // all live lanes have arrived: open the next phase
        arrived = 0;
        bal[(phase & 1) ^ 1] = 0;  // everyone has consumed the phase before this one
        phase++;
    }
};
struct BlockSync {
    int alive = 0, arrived = 0;
    uint64_t phase = 0;
};
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    Idx tid;
    int lane = 0;
    WaveSync *wave = nullptr;
    bool done = false;
    const uint64_t *wait_ptr = nullptr;
    uint64_t wait_val = 0;
};

Fiber *cur = nullptr;
Idx cur_tid, cur_bid, cur_bdim, cur_gdim;
static ucontext_t sched_ctx;
static BlockSync blk_single;
static BlockSync *blkp = &blk_single;  // the running workgroup's barrier state
static const std::function<void()> *cur_body;
static const size_t STACK = 256 * 1024;

static unsigned long long n_waits = 0, n_yields = 0;
static struct Report { ~Report() { if (getenv("MRZ_EMU_COUNT")) fprintf(stderr, "emu: %llu waits, %llu yields\n", n_waits, n_yields); } } report;
static void yield_wait(const uint64_t *ptr, uint64_t val) {
    n_waits++;
    Fiber *me = cur;
    me->wait_ptr = ptr;
    me->wait_val = val;
    swapcontext(&me->ctx, &sched_ctx);
    // resumed: scheduler restored cur / cur_tid
}

static void wave_arrive(WaveSync *w) {
    if (++w->arrived >= w->alive) {
        w->complete();
    } else {
        const uint64_t ph = w->phase;
        yield_wait(&w->phase, ph);
    }
}

unsigned long long ballot(int pred) {
    WaveSync *w = cur->wave;
    const int par = (int)(w->phase & 1);
    if (pred) w->bal[par] |= 1ull << cur->lane;
    wave_arrive(w);
    return w->bal[par];
}

int shfl(int v, int src) {
    WaveSync *w = cur->wave;
    const int par = (int)(w->phase & 1);
    w->slot[par][cur->lane] = (int64_t)(uint32_t)v;
    wave_arrive(w);
    return (int)(uint32_t)w->slot[par][src & 63];
}

int shfl_xor(int v, int mask) {
    WaveSync *w = cur->wave;
    const int par = (int)(w->phase & 1);
    w->slot[par][cur->lane] = (int64_t)(uint32_t)v;
    const int src = (cur->lane ^ mask) & 63;
    wave_arrive(w);
    return (int)(uint32_t)w->slot[par][src];
}

// a spin-wait's sleep: let the other fibers run
void yield() {
    n_yields++;
    Fiber *me = cur;
    me->wait_ptr = nullptr;
    swapcontext(&me->ctx, &sched_ctx);
}

int first_live_lane() { return __builtin_ffsll((long long)cur->wave->live_mask) - 1; }

void syncthreads() {
    if (++blkp->arrived >= blkp->alive) {
        blkp->arrived = 0;
        blkp->phase++;
    } else {
        const uint64_t ph = blkp->phase;
        yield_wait(&blkp->phase, ph);
    }
}

static void fiber_main() {
    (*cur_body)();
    Fiber *me = cur;
    me->done = true;
    // a finished thread no longer takes part in rendezvous
    WaveSync *w = me->wave;
    w->alive--;
    w->live_mask &= ~(1ull << me->lane);
    if (w->alive > 0 && w->arrived >= w->alive) w->complete();
    blkp->alive--;
    if (blkp->alive > 0 && blkp->arrived >= blkp->alive) {
        blkp->arrived = 0;
        blkp->phase++;
    }
    swapcontext(&me->ctx, &sched_ctx);
}

// ---- co-resident mode ----------------------------------------------------------------------------------------------
// MRZ_EMU_CORESIDENT=1 in the environment: the workgroups of a (small) grid run interleaved instead of one after
// another, so that kernels whose workgroups wait for each other (the sequencer's token) can be executed.  Still one
// OS thread: a workgroup runs until every one of its fibers is waiting or has yielded, then the next one does.
// `__shared__` is a function-local static here, i.e. ONE copy for all workgroups: a kernel that is to run in this mode
// indexes its LDS by workgroup under MRZ_EMU_LDS_PER_BLOCK.
struct BlockCtx {
    Idx bid;
    BlockSync blk;
    std::vector<WaveSync> waves;
    std::vector<Fiber> fibers;
    unsigned remaining = 0;
};

static bool coresident_requested = false;
void request_coresident() { coresident_requested = true; }  // the next launch is of a kernel written for it

static void launch_coresident(dim3 grid, dim3 block, const std::function<void()> &body) {
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned nwaves = (nthreads + 63) / 64;
    const unsigned nblocks = grid.x * grid.y * grid.z;
    static std::vector<BlockCtx *> pool;
    while (pool.size() < nblocks) pool.push_back(new BlockCtx());
    cur_body = &body;
    cur_bdim = { block.x, block.y, block.z };
    cur_gdim = { grid.x, grid.y, grid.z };
    for (unsigned b = 0; b < nblocks; b++) {
        BlockCtx &B = *pool[b];
        B.bid = { b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y) };
        B.blk = BlockSync();
        B.blk.alive = (int)nthreads;
        if (B.fibers.size() < nthreads) {
            size_t old = B.fibers.size();
            B.fibers.resize(nthreads);
            for (size_t i = old; i < nthreads; i++) B.fibers[i].stack = (char *)malloc(STACK);
        }
        if (B.waves.size() < nwaves) B.waves.resize(nwaves);
        for (unsigned wv = 0; wv < nwaves; wv++) {
            B.waves[wv] = WaveSync();
            unsigned lanes = nthreads - wv * 64;
            B.waves[wv].alive = lanes > 64 ? 64 : (int)lanes;
            B.waves[wv].live_mask = B.waves[wv].alive == 64 ? ~0ull : ((1ull << B.waves[wv].alive) - 1);
            memset(B.waves[wv].slot, 0, sizeof(B.waves[wv].slot));
        }
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber &f = B.fibers[t];
            f.tid = { t % block.x, (t / block.x) % block.y, t / (block.x * block.y) };
            f.lane = (int)(t & 63);
            f.wave = &B.waves[t / 64];
            f.done = false;
            f.wait_ptr = nullptr;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = &sched_ctx;
            makecontext(&f.ctx, fiber_main, 0);
        }
        B.remaining = nthreads;
    }
    unsigned live_blocks = nblocks;
    unsigned long long idle_rounds = 0;
    while (live_blocks) {
        bool any = false;
        for (unsigned b = 0; b < nblocks; b++) {
            BlockCtx &B = *pool[b];
            if (!B.remaining) continue;
            // a few passes over this workgroup, then the next one gets the processor
            for (int pass = 0; pass < 4 && B.remaining; pass++) {
                bool progressed = false;
                for (unsigned t = 0; t < nthreads; t++) {
                    Fiber &f = B.fibers[t];
                    if (f.done) continue;
                    if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
                    f.wait_ptr = nullptr;
                    cur = &f;
                    cur_tid = f.tid;
                    cur_bid = B.bid;
                    blkp = &B.blk;
                    swapcontext(&sched_ctx, &f.ctx);
                    progressed = true;
                    if (f.done) B.remaining--;
                }
                if (!progressed) break;
                any = true;
            }
            if (!B.remaining) live_blocks--;
        }
        if (!any && ++idle_rounds > 4) {
            fprintf(stderr, "emu: deadlock in co-resident grid\n");
            abort();
        }
    }
    cur = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    // one kernel at a time: the scheduler's state is global, and host threads of the library (the pipeline's
    // consumer runs the LZ4 gate while the producer sequences) may launch concurrently
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    const bool asked = coresident_requested;
    coresident_requested = false;
    {
        const char *e = getenv("MRZ_EMU_CORESIDENT");
        if (asked && e && *e == '1' && grid.x * grid.y * grid.z > 1 && grid.x * grid.y * grid.z <= 64) {
            launch_coresident(grid, block, body);
            return;
        }
    }
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned nwaves = (nthreads + 63) / 64;
    static std::vector<Fiber> fibers;
    static std::vector<WaveSync> waves;
    if (fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) fibers[i].stack = (char *)malloc(STACK);
    }
    if (waves.size() < nwaves) waves.resize(nwaves);
    cur_body = &body;
    cur_bdim = { block.x, block.y, block.z };
    cur_gdim = { grid.x, grid.y, grid.z };
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                cur_bid = { bx, by, bz };
                blkp = &blk_single;
                blk_single = BlockSync();
                blk_single.alive = (int)nthreads;
                for (unsigned wv = 0; wv < nwaves; wv++) {
                    waves[wv] = WaveSync();
                    unsigned lanes = nthreads - wv * 64;
                    waves[wv].alive = lanes > 64 ? 64 : (int)lanes;
                    waves[wv].live_mask = waves[wv].alive == 64 ? ~0ull : ((1ull << waves[wv].alive) - 1);
                    memset(waves[wv].slot, 0, sizeof(waves[wv].slot));
                }
                for (unsigned t = 0; t < nthreads; t++) {
                    Fiber &f = fibers[t];
                    f.tid = { t % block.x, (t / block.x) % block.y, t / (block.x * block.y) };
                    f.lane = (int)(t & 63);
                    f.wave = &waves[t / 64];
                    f.done = false;
                    f.wait_ptr = nullptr;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = &sched_ctx;
                    makecontext(&f.ctx, fiber_main, 0);
                }
                unsigned remaining = nthreads;
                while (remaining) {
                    bool progressed = false;
                    for (unsigned t = 0; t < nthreads; t++) {
                        Fiber &f = fibers[t];
                        if (f.done) continue;
                        if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
                        f.wait_ptr = nullptr;
                        cur = &f;
                        cur_tid = f.tid;
                        swapcontext(&sched_ctx, &f.ctx);
                        progressed = true;
                        if (f.done) remaining--;
                    }
                    if (!progressed) {
                        fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent collective?\n", bx, by, bz);
                        abort();
                    }
                }
            }
    cur = nullptr;
}

}  // namespace emu
