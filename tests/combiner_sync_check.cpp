// Stress of the bsg_query combiner's synchronisation on the CPU (bloomsearch_amd/csrc/host/combiner_sync.hpp: the very code the
// library instantiates), built with plain g++ — under ThreadSanitizer where the toolchain has it — by tests/test_combiner_sync.py.
// T threads submit calls through one Gate; whoever collects serves its cycle with a stand-in for the device (a short busy wait, then
// out = f(in) for every call of the cycle).  Checked: every call returns served exactly once with ITS answer, cycles really combine,
// the gate ends idle (no role held, no slot taken, nothing left on the stacks), nothing deadlocks (a watchdog aborts).
//   usage: combiner_sync_check <threads> <calls per thread> <max_inflight> <spin_us> <work_ns>
#include "combiner_sync.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <csignal>

struct Req : bsgsync::Waiter {
    uint64_t in = 0, out = 0;
    uint32_t served = 0;
};

static uint64_t f(uint64_t x) { return x * 0x9E3779B97F4A7C15ull + 12345; }

int main(int argc, char **argv)
{
    const uint32_t T = argc > 1 ? atoi(argv[1]) : 16, per = argc > 2 ? atoi(argv[2]) : 2000;
    bsgsync::Gate gate;
    gate.max_inflight = argc > 3 ? atoi(argv[3]) : 2;
    gate.spin_us = argc > 4 ? atoi(argv[4]) : 60;
    const uint64_t work_ns = argc > 5 ? strtoull(argv[5], nullptr, 10) : 5000;
    signal(SIGALRM, [](int) { fprintf(stderr, "watchdog: the combiner did not finish in time (deadlock?)\n"); _exit(3); });
    alarm(120);
    std::atomic<uint64_t> cycles{0}, cycle_calls{0}, max_cycle{0}, bad{0}, inflight_now{0}, inflight_max{0};
    auto run_cycle = [&](Req &me) {
        std::vector<Req *> cyc;
        bsgsync::drain(gate, me, cyc);
        const uint64_t now = inflight_now.fetch_add(1) + 1;
        uint64_t m = inflight_max.load();
        while (m < now && !inflight_max.compare_exchange_weak(m, now)) {}
        bsgsync::release_role(gate);              // the next cycle is collected while this one "runs"
        const auto t0 = std::chrono::steady_clock::now();
        while ((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() < work_ns) {}
        for (Req *r : cyc) { r->out = f(r->in); r->served++; }
        cycles++; cycle_calls += cyc.size();
        uint64_t mc = max_cycle.load();
        while (mc < cyc.size() && !max_cycle.compare_exchange_weak(mc, cyc.size())) {}
        inflight_now.fetch_sub(1);
        bsgsync::release_slot(gate);
        bsgsync::release_cycle(cyc);
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            for (uint32_t i = 0; i < per; ++i) {
                Req r;
                r.in = ((uint64_t)t << 32) | i;
                bsgsync::submit(gate, r, run_cycle);
                if (r.out != f(r.in) || r.served != 1) bad++;
            }
        });
    for (auto &x : th) x.join();
    const bool idle = gate.gate.load() == 0 && bsgsync::waiting_stack(gate) == bsgsync::kStacks;
    printf("threads %u calls %llu cycles %llu max_cycle %llu inflight_max %llu bad %llu idle %d\n", T, (unsigned long long)cycle_calls.load(),
           (unsigned long long)cycles.load(), (unsigned long long)max_cycle.load(), (unsigned long long)inflight_max.load(), (unsigned long long)bad.load(), idle ? 1 : 0);
    const uint64_t limit = gate.max_inflight + 1;
    return (bad.load() == 0 && idle && cycle_calls.load() == (uint64_t)T * per && inflight_max.load() <= limit) ? 0 : 1;
}
