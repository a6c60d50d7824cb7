// hipemu.cpp -- TEST INFRASTRUCTURE ONLY (see hipemu.h): fiber scheduler of the host-side HIP emulation.
#include "hipemu.h"

#include <vector>

namespace hipemu {

Block *g_blk = nullptr;
Fiber *g_cur = nullptr;

namespace {
constexpr size_t STACK = 256 << 10;
std::vector<char *> g_stacks;
const std::function<void()> *g_body = nullptr;

void trampoline() {
    (*g_body)();
    g_cur->done = true;
    g_blk->progress = true;
    swapcontext(&g_cur->ctx, &g_blk->sched);
}

[[noreturn]] void deadlock(const char *why) {
    fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %s\n", g_blk->bid.x, g_blk->bid.y, g_blk->bid.z, why);
    for (int t = 0; t < g_blk->nthreads; t++) {
        const Fiber &f = g_blk->fib[t];
        if (!f.done) fprintf(stderr, "  thread %d (wave %d lane %d): barriers passed %llu, wave ops %llu\n", t, f.wave, f.lane,
                             (unsigned long long)f.bar_count, (unsigned long long)f.op_count);
    }
    abort();
}
}  // namespace

void yield() { swapcontext(&g_cur->ctx, &g_blk->sched); }

void syncthreads() {
    Fiber *me = g_cur;
    const uint64_t target = ++me->bar_count;
    g_blk->progress = true;          // arriving is progress
    for (;;) {
        bool all = true;
        for (int t = 0; t < g_blk->nthreads; t++) {
            const Fiber &f = g_blk->fib[t];
            if (!f.done && f.bar_count < target) { all = false; break; }
        }
        if (all) return;
        yield();
    }
}

void wave_xchg(const void *val, size_t sz, unsigned char (*all)[16], uint64_t *mask) {
    Fiber *me = g_cur;
    Block *b = g_blk;
    const uint64_t k = ++me->op_count;
    WaveBox &box = b->waves[me->wave];
    memset(box.buf[k & 1][me->lane], 0, 16);
    memcpy(box.buf[k & 1][me->lane], val, sz);
    b->progress = true;
    const int t0 = me->wave * 64, t1 = std::min(b->nthreads, t0 + 64);
    for (;;) {
        bool ready = true;
        for (int t = t0; t < t1; t++) {
            const Fiber &f = b->fib[t];
            if (!f.done && f.op_count < k) { ready = false; break; }
        }
        if (ready) break;
        yield();
    }
    uint64_t m = 0;
    for (int t = t0; t < t1; t++) {
        const Fiber &f = b->fib[t];
        if (f.op_count >= k) { m |= 1ull << (t - t0); memcpy(all[t - t0], box.buf[k & 1][t - t0], 16); }
        else memset(all[t - t0], 0, 16);
    }
    for (int l = t1 - t0; l < 64; l++) memset(all[l], 0, 16);
    *mask = m;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", nt); abort(); }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
    while ((int)g_stacks.size() < nt) g_stacks.push_back((char *)malloc(STACK));
    Block blk;
    blk.bdim = block; blk.gdim = grid; blk.nthreads = nt;
    std::vector<Fiber> fibers((size_t)nt);
    std::vector<WaveBox> waves((size_t)((nt + 63) / 64));
    blk.fib = fibers.data(); blk.waves = waves.data();
    Block *prev_blk = g_blk; Fiber *prev_cur = g_cur; const std::function<void()> *prev_body = g_body;
    g_blk = &blk; g_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blk.bid = dim3(bx, by, bz);
                for (int t = 0; t < nt; t++) {
                    Fiber &f = fibers[(size_t)t];
                    f.tid = dim3((unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y));
                    f.lane = t & 63; f.wave = t >> 6; f.done = false; f.bar_count = 0; f.op_count = 0;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = g_stacks[(size_t)t]; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
                int left = nt;
                while (left > 0) {
                    blk.progress = false;
                    left = 0;
                    for (int t = 0; t < nt; t++) {
                        Fiber &f = fibers[(size_t)t];
                        if (f.done) continue;
                        g_cur = &f;
                        swapcontext(&blk.sched, &f.ctx);
                        if (!f.done) left++;
                    }
                    if (left > 0 && !blk.progress) deadlock("no thread can make progress (divergent barrier or wave operation)");
                }
            }
    g_blk = prev_blk; g_cur = prev_cur; g_body = prev_body;
}

}  // namespace hipemu
