#include "hipemu.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <algorithm>
#include <deque>
#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

// Fibre switch.  glibc's swapcontext saves / restores the signal mask with a system call on every switch, and a lock-step emulation of
// 64-wide shuffles switches millions of times per test; on x86-64 the switch is therefore the six callee-saved registers + the stack
// pointer, by hand (every fibre runs the same code with the same floating-point control state).  Other hosts keep ucontext.
#if defined(__x86_64__)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");
#else
#define HIPEMU_FAST_SWITCH 0
#endif

namespace hipemu {
void dma_wait(int max_outstanding);
namespace {
struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;            // fast switch: the fibre's saved stack pointer
    char* stack = nullptr;
    bool done = false;
    bool waiting = false;
    dim3 tid;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
ucontext_t main_ctx;
void* main_sp = nullptr;
int cur = -1;
const std::function<void()>* body_fn = nullptr;
int bar_count = 0;
unsigned bar_gen = 0;
std::vector<int> wave_count;
std::vector<unsigned> wave_gen;
std::vector<char> wave_scratch;
std::vector<char> smem;
struct PendingDma {
    char* dst;
    unsigned char data[16];
    int n;
};
std::vector<std::deque<PendingDma>> dma_q;          // per thread, in issue order
const bool dma_sync = getenv("HIPEMU_SYNC_DMA") != nullptr;
int n_threads = 0;
unsigned long ticks = 0;     // bumped on every barrier arrival / thread exit: the scheduler's progress signal

void yield() {
    Fiber& f = fibers[cur];
#if HIPEMU_FAST_SWITCH
    hipemu_switch(&f.sp, main_sp);
#else
    swapcontext(&f.ctx, &main_ctx);
#endif
}
void trampoline() {
    (*body_fn)();
    dma_wait(0);                               // the end of the program retires everything
    fibers[cur].done = true;
#if HIPEMU_FAST_SWITCH
    hipemu_switch(&fibers[cur].sp, main_sp);
    abort();                                   // a finished fibre is never resumed
#else
    swapcontext(&fibers[cur].ctx, &main_ctx);
#endif
}
#if HIPEMU_FAST_SWITCH
// first activation: hipemu_switch pops six zeroed registers and "returns" into trampoline with the stack aligned as after a call
void prepare(Fiber& f) {
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** x = (void**)(top - 8);              // rsp after the ret: 8 mod 16, holds a null return address for trampoline
    x[0] = nullptr;
    x[-1] = (void*)&trampoline;
    for (int i = 2; i <= 7; ++i) x[-i] = nullptr;
    f.sp = (void*)(x - 7);
}
#endif
}  // namespace

void dma_issue(void* lds_dst, const void* src, int bytes) {
    if (dma_sync) {
        if (src) memcpy(lds_dst, src, bytes); else memset(lds_dst, 0, bytes);
        return;
    }
    PendingDma p;
    p.dst = (char*)lds_dst;
    p.n = bytes;
    if (src) memcpy(p.data, src, bytes); else memset(p.data, 0, bytes);
    memset(lds_dst, 0xFF, bytes);
    dma_q[cur].push_back(p);
}
void dma_wait(int max_outstanding) {
    auto& q = dma_q[cur];
    while ((int)q.size() > max_outstanding) {
        memcpy(q.front().dst, q.front().data, q.front().n);
        q.pop_front();
    }
}

char* dyn_smem() { return smem.data(); }
void* wave_buf() { return wave_scratch.data() + (size_t)(cur / 64) * 64 * 256; }

void wave_sync() {
    const int w = cur / 64;
    const int lanes = std::min(64, n_threads - w * 64);
    const unsigned gen = wave_gen[w];
    ++ticks;
    if (++wave_count[w] == lanes) {
        wave_count[w] = 0;
        wave_gen[w]++;
    } else {
        fibers[cur].waiting = true;
        while (wave_gen[w] == gen) yield();
        fibers[cur].waiting = false;
    }
}

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
    n_threads = (int)(block.x * block.y * block.z);
    gridDim = grid;
    blockDim = block;
    body_fn = &body;
    smem.assign(dyn_smem_bytes + 64, 0);
    wave_count.assign((n_threads + 63) / 64, 0);
    wave_gen.assign((n_threads + 63) / 64, 0);
    wave_scratch.assign((size_t)((n_threads + 63) / 64) * 64 * 256, 0);
    dma_q.assign(n_threads, {});
    if ((int)fibers.size() < n_threads) {
        size_t old = fibers.size();
        fibers.resize(n_threads);
        for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char*)malloc(kStack);
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                bar_count = 0;
                for (auto& c : wave_count) c = 0;
                for (auto& q : dma_q) q.clear();
                for (int t = 0; t < n_threads; ++t) {
                    Fiber& f = fibers[t];
                    f.done = false;
                    f.waiting = false;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
#if HIPEMU_FAST_SWITCH
                    prepare(f);
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &main_ctx;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
                }
                int alive = n_threads;
                long spins = 0;
                // HIPEMU_ORDER: the order in which the WAVES of a workgroup get the processor between rendezvous points.  Between two barriers the
                // hardware may run waves in any order, so a correct kernel gives the same bits under every choice; "reverse" and "random[:seed]"
                // expose an LDS hand-over between waves that has no barrier (with the default ascending order the producer wave of such a
                // hand-over often happens to run first).  Lanes of a wave stay in ascending order (wave-collective operations rendezvous anyway).
                static const char* order_env = getenv("HIPEMU_ORDER");
                const int n_waves = (n_threads + 63) / 64;
                std::vector<int> wave_order(n_waves);
                for (int w = 0; w < n_waves; ++w) wave_order[w] = w;
                const bool order_random = order_env && !strncmp(order_env, "random", 6);
                if (order_env && !strcmp(order_env, "reverse"))
                    for (int w = 0; w < n_waves; ++w) wave_order[w] = n_waves - 1 - w;
                static unsigned long rng = order_random && strchr(order_env, ':') ? strtoul(strchr(order_env, ':') + 1, nullptr, 10) * 2654435761ul + 1 : 88172645463325252ul;
                while (alive > 0) {
                    const unsigned long before = ticks;
                    if (order_random)
                        for (int w = n_waves - 1; w > 0; --w) {
                            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                            std::swap(wave_order[w], wave_order[rng % (unsigned long)(w + 1)]);
                        }
                    for (int idx = 0; idx < n_waves * 64; ++idx) {
                        const int t = wave_order[idx / 64] * 64 + idx % 64;
                        if (t >= n_threads) continue;
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        cur = t;
                        threadIdx = f.tid;
#if HIPEMU_FAST_SWITCH
                        hipemu_switch(&main_sp, f.sp);
#else
                        swapcontext(&main_ctx, &f.ctx);
#endif
                        if (f.done) { --alive; ++ticks; }
                    }
                    if (ticks == before) {
                        if (++spins > 2) {
                            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d threads stuck at a barrier\n", bx, by, bz, alive);
                            abort();
                        }
                    } else {
                        spins = 0;
                    }
                }
            }
    cur = -1;
}
}  // namespace hipemu

namespace hipemu {
void barrier() {
    const unsigned gen = bar_gen;
    ++ticks;
    if (++bar_count == n_threads) {
        bar_count = 0;
        bar_gen++;
    } else {
        fibers[cur].waiting = true;
        while (bar_gen == gen) yield();
        fibers[cur].waiting = false;
    }
}
}  // namespace hipemu

// hipcc's __syncthreads() is fence + s_barrier: s_waitcnt vmcnt(0) lgkmcnt(0) first — outstanding LDS-DMA pieces of the thread are retired
void __syncthreads() {
    hipemu::dma_wait(0);
    hipemu::barrier();
}
