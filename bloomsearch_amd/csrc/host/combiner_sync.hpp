// combiner_sync.hpp — the synchronisation of the bsg_query combiner (csrc/combine_api.inc), free of any device type: who collects,
// how calls wait, how a cycle's callers are released.  The library instantiates it over its request type; tests/combiner_sync_check.cpp
// runs the same code on the CPU, many threads against a stand-in cycle, under ThreadSanitizer (tests/test_combiner_sync.py).
//
// Protocol (one Gate per context):
//   * a call PUSHES its Waiter on one of kStacks lock-free stacks (callers push; only the holder of the collector role takes), then
//     tries for the role + a cycle slot; whoever gives the role or a slot back looks at the stacks once more and, if calls wait there,
//     takes role + slot again ON BEHALF of the newest arrival (the call most likely still polling) and appoints it.  Push-then-try on
//     one side, release-then-look on the other, all sequentially consistent: a call can never wait with the role free and a slot open.
//   * a collector DRAINS the stacks (drain), gives the role back at once (release_role: the next cycle is collected while this one is
//     prepared and run), serves its cycle, gives the slot back (release_slot) and RELEASES the cycle's callers (release_cycle).
//   * a waiting call polls briefly while the context is quiet, else sleeps in a futex on its own state word.
//   * release is a TREE over the cycle's calls (call i wakes calls 4i + 1 .. 4i + 4; the collector is call 0) and nothing of it is
//     shared: the collector marks the calls served from the LAST to the first, and before it marks call i it writes into call i's
//     Waiter which of its children (all marked already) were asleep.  The waker's last access to a Waiter is the exchange itself — the
//     Waiter lives on its caller's stack and may be gone the moment the new state is seen; a futex_wake afterwards only names the address.
#pragma once
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace bsgsync {

struct Waiter {
    Waiter *next = nullptr;                           // the stack of waiting calls
    std::atomic<uint32_t> state{0};                   // 0 queued, 3 queued + asleep, 1 appointed collector, 2 served
    std::atomic<uint32_t> *child[4] = {nullptr, nullptr, nullptr, nullptr};   // whom this caller wakes once it is served
    uint32_t n_child = 0;
};

constexpr uint32_t kStacks = 8;                       // with 256 callers ONE stack head is a cache line every call rewrites
constexpr uint32_t kWakeFan = 4;
constexpr uint32_t kGateCollecting = 1u << 31;        // gate bit 31: the collector role is taken; low bits: cycles in flight

struct Gate {
    struct alignas(64) Head { std::atomic<Waiter *> p{nullptr}; };
    Head head[kStacks];
    alignas(64) std::atomic<uint32_t> gate{0};
    std::atomic<uint32_t> mean_cycle_x16{0};          // running mean of the cycles' sizes, x 16: how busy the context is
    uint32_t max_inflight = 2;                        // cycles in flight; one more while cycles average > 32 calls
    uint32_t spin_us = 60;                            // a queued caller polls this long while the context is quiet (cycles of <= 8 calls)
};

inline void futex_wait(std::atomic<uint32_t> *a, uint32_t v) { (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAIT_PRIVATE, v, nullptr, nullptr, 0); }
inline void futex_wake(std::atomic<uint32_t> *a) { (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

inline void set_state(Waiter &w, uint32_t v)
{
    std::atomic<uint32_t> *addr = &w.state;
    if (addr->exchange(v, std::memory_order_seq_cst) == 3u) futex_wake(addr);
}

inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// Poll briefly (a cycle on an idle device is tens of microseconds: the answer often arrives while polling), then sleep in a futex:
// with as many callers as processors, callers that keep polling take the processor from the collector and the runtime's threads.
inline void wait_while_queued(Waiter &w, uint32_t spin_us)
{
    if (spin_us) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0;; ++i) {
            if (w.state.load(std::memory_order_acquire) != 0) return;
            cpu_relax();
            if ((i & 31) == 31 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
    }
    uint32_t expect = 0;
    if (!w.state.compare_exchange_strong(expect, 3u, std::memory_order_seq_cst)) return;      // served / appointed meanwhile
    while (w.state.load(std::memory_order_seq_cst) == 3u) futex_wait(&w.state, 3u);
}

inline void wake_children(Waiter &w)
{
    for (uint32_t k = 0; k < w.n_child; ++k) futex_wake(w.child[k]);
}

// cyc[0] is the collector itself (never marked: it is running)
template <class W>
inline void release_cycle(std::vector<W *> &cyc)
{
    const uint32_t n = (uint32_t)cyc.size();
    std::vector<std::atomic<uint32_t> *> addr(n);
    std::vector<uint8_t> asleep(n, 0);
    for (uint32_t i = 0; i < n; ++i) addr[i] = &static_cast<Waiter *>(cyc[i])->state;
    for (uint32_t i = n; i-- > 0;) {
        Waiter &w = *static_cast<Waiter *>(cyc[i]);
        w.n_child = 0;
        for (uint64_t ch = (uint64_t)i * kWakeFan + 1; ch <= (uint64_t)i * kWakeFan + kWakeFan && ch < n; ++ch)
            if (asleep[ch]) w.child[w.n_child++] = addr[ch];
        // (from here on call i may be gone the moment its state is seen: only its address is named afterwards)
        if (i > 0 && addr[i]->exchange(2u, std::memory_order_seq_cst) == 3u) asleep[i] = 1;
    }
    wake_children(*static_cast<Waiter *>(cyc[0]));
}

inline bool try_collect(Gate &g)
{
    uint32_t v = g.gate.load(std::memory_order_seq_cst);
    for (;;) {
        // one more cycle in flight while the context is busy (cycles of > 32 calls on average: 1.26 -> 1.65 M calls/s at 256 callers);
        // with fewer callers a third cycle only splits what two would have carried (64 callers: 8.6 -> 7.0 x 10^5; 16: 3.5 -> 2.5)
        const uint32_t limit = g.max_inflight + (g.mean_cycle_x16.load(std::memory_order_relaxed) > 32u * 16u ? 1u : 0u);
        if ((v & kGateCollecting) || (v & 0xFFFFu) >= limit) return false;
        if (g.gate.compare_exchange_weak(v, (v | kGateCollecting) + 1u, std::memory_order_seq_cst)) return true;
    }
}

inline uint32_t waiting_stack(Gate &g)       // the first stack that holds a call, or kStacks
{
    for (uint32_t k = 0; k < kStacks; ++k) if (g.head[k].p.load(std::memory_order_seq_cst) != nullptr) return k;
    return kStacks;
}

// appoint the newest call of some stack if calls wait, the role is free and a slot is
inline void kick(Gate &g)
{
    for (;;) {
        if (waiting_stack(g) == kStacks || !try_collect(g)) return;
        // pop (single consumer: only the holder of the role takes from the stacks, so there is no ABA)
        const uint32_t k = waiting_stack(g);
        Waiter *t = k < kStacks ? g.head[k].p.load(std::memory_order_acquire) : nullptr;
        while (t && !g.head[k].p.compare_exchange_weak(t, t->next, std::memory_order_acq_rel)) {}
        if (t) { set_state(*t, 1); return; }
        g.gate.fetch_and(~kGateCollecting, std::memory_order_seq_cst);     // the stacks were emptied between waiting_stack and here (a collector's drain took them): give both back and look again
        g.gate.fetch_sub(1u, std::memory_order_seq_cst);
    }
}

inline void release_role(Gate &g) { g.gate.fetch_and(~kGateCollecting, std::memory_order_seq_cst); kick(g); }
inline void release_slot(Gate &g) { g.gate.fetch_sub(1u, std::memory_order_seq_cst); kick(g); }

// The collector takes everything that waits — itself in front, then every stack's calls oldest first — and notes how busy the context is.
template <class W>
inline void drain(Gate &g, W &me, std::vector<W *> &cyc)
{
    cyc.clear();
    cyc.push_back(&me);
    for (uint32_t k = 0; k < kStacks; ++k) {
        const size_t first = cyc.size();
        for (Waiter *w = g.head[k].p.exchange(nullptr, std::memory_order_acq_rel); w; w = w->next)
            if (w != static_cast<Waiter *>(&me)) cyc.push_back(static_cast<W *>(w));
        std::reverse(cyc.begin() + first, cyc.end());
    }
    const uint32_t m = g.mean_cycle_x16.load(std::memory_order_relaxed);
    g.mean_cycle_x16.store(m - m / 8 + 2 * (uint32_t)std::min<size_t>(cyc.size(), 4096), std::memory_order_relaxed);
}

struct PhaseClock { uint64_t wait = 0, duty = 0, collect = 0; uint64_t (*now)() = nullptr; };      // lab: a caller's own processor time per phase

// One call through the combiner.  run_cycle(me): called on the thread that collects, with role + a slot held; it must drain(), then
// release_role(), serve every call of its cycle, release_slot() and release_cycle().
template <class W, class RunCycle>
inline void submit(Gate &g, W &w, RunCycle &&run_cycle, PhaseClock *pc = nullptr)
{
    Waiter &me = w;
    const uint64_t t0 = pc ? pc->now() : 0;
    std::atomic<Waiter *> &head = g.head[((uintptr_t)&me >> 14) % kStacks].p;      // (a caller's stack frame: threads land on different stacks)
    me.next = head.load(std::memory_order_relaxed);
    while (!head.compare_exchange_weak(me.next, &me, std::memory_order_seq_cst)) {}
    if (try_collect(g)) {
        // (the call may have been taken into another collector's cycle between the push and the role: then it is being served, and
        // role + slot go straight back)
        bool mine = false;
        for (Waiter *q = head.load(std::memory_order_acquire); q; q = q->next) if (q == &me) { mine = true; break; }
        if (mine) { run_cycle(w); if (pc) pc->collect += pc->now() - t0; return; }
        g.gate.fetch_sub(1u, std::memory_order_seq_cst);
        release_role(g);
    }
    // polling is worth its processor time only while the context is quiet (the answer is tens of microseconds away and a futex round
    // trip costs about as much); with many callers, polling them all starves the collector
    wait_while_queued(me, g.mean_cycle_x16.load(std::memory_order_relaxed) <= 8 * 16 ? g.spin_us : 0u);
    const uint64_t t1 = pc ? pc->now() : 0;
    if (pc) pc->wait += t1 - t0;
    if (me.state.load(std::memory_order_acquire) == 1) { run_cycle(w); if (pc) pc->collect += pc->now() - t1; }
    else { wake_children(me); if (pc) pc->duty += pc->now() - t1; }      // served by somebody else's cycle: pass the wake-up on
}

}  // namespace bsgsync
