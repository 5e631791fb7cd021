// conc_driver.cpp — T host threads driving bsg_query on ONE context, the way the reference's file workers would (one goroutine
// per candidate file, several Query() calls at once: query_exec.go:303-357, 427-431).  Test and bench harness over the C-ABI
// (include/bloomgpu.h); nothing of the product lives here.  Python threads cannot play this part: the interpreter lock serialises
// the marshalling around every call, so at 256 threads the GIL, not the library, would set the rate.
//
//   conc_run(ctx, n_threads, seconds, queries..., arenas..., expected..., out...)
// Every thread loops: pick a query (round-robin from its own offset) and `arenas_per_call` consecutive arena ids (rotating through
// the list), call bsg_query, compare the survivors with the expected rows (every arena of the list holds the same filters), record
// the latency.  Returns the calls made; mismatches and errors are counted.
#include "bloomgpu.h"

#include <atomic>
#include <chrono>
#include <ctime>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

struct conc_query {            // one query = one bsg_query call's term / program arguments (n_queries = 1)
    const uint8_t *term_bytes;
    const uint32_t *term_off;
    const uint32_t *term_kinds;
    uint32_t n_terms;
    const uint32_t *prog_ops;
    uint32_t n_ops;
};

struct conc_result {
    uint64_t calls;
    uint64_t mismatches;
    uint64_t errors;
    double seconds;            // wall time of the measured window
    uint64_t n_lat;            // latencies recorded (<= lat_cap)
    double cpu_seconds;        // processor time the whole process used in the window (callers polling / sleeping / collecting, runtime threads)
};

// expected: [n_queries][G] words (G = ceil(n_blocks / 64)); lat_ns: room for lat_cap samples (every `lat_stride`-th call of thread 0..)
__attribute__((visibility("default")))
int32_t conc_run(bsg_ctx *ctx, uint32_t n_threads, double seconds, const conc_query *queries, uint32_t n_queries,
                 const uint64_t *arena_ids, uint32_t n_arena_ids, uint32_t arenas_per_call, uint32_t n_blocks, const uint64_t *expected,
                 uint64_t *lat_ns, uint64_t lat_cap, conc_result *out)
{
    const uint32_t G = (n_blocks + 63) / 64;
    std::atomic<uint64_t> calls{0}, mismatches{0}, errors{0};
    std::atomic<uint32_t> ready{0};
    std::atomic<int> go{0};
    std::vector<std::thread> th;
    const uint64_t per_thread = lat_cap / n_threads;
    std::vector<uint64_t> recorded(n_threads, 0);
    auto body = [&](uint32_t tid) {
        bsg_ctx *scope = nullptr;
        if (bsg_scope_open(ctx, &scope) != BSG_OK) { errors++; scope = ctx; }
        std::vector<uint64_t> ids(arenas_per_call), got((size_t)arenas_per_call * G);
        uint32_t qi = tid % n_queries, ai = (tid * arenas_per_call) % n_arena_ids;
        const uint32_t poff[2] = {0, 0};
        {   // one untimed call first: a thread's first call pays the runtime's per-thread setup (milliseconds when 256 threads start at once)
            const conc_query &q = queries[qi];
            for (uint32_t j = 0; j < arenas_per_call; ++j) ids[j] = arena_ids[(ai + j) % n_arena_ids];
            uint32_t off2[2] = {0, q.n_ops};
            if (bsg_query(scope, ids.data(), arenas_per_call, q.term_bytes, q.term_off, q.term_kinds, q.n_terms, q.prog_ops, off2, 1, got.data()) != BSG_OK) errors++;
        }
        ready++;
        while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        uint64_t mine = 0;
        while (go.load(std::memory_order_relaxed) == 1) {
            const conc_query &q = queries[qi];
            for (uint32_t j = 0; j < arenas_per_call; ++j) ids[j] = arena_ids[(ai + j) % n_arena_ids];
            uint32_t off2[2] = {poff[0], q.n_ops};
            const auto t0 = std::chrono::steady_clock::now();
            const int32_t rc = bsg_query(scope, ids.data(), arenas_per_call, q.term_bytes, q.term_off, q.term_kinds, q.n_terms, q.prog_ops, off2, 1, got.data());
            const auto t1 = std::chrono::steady_clock::now();
            if (rc != BSG_OK) errors++;
            else
                for (uint32_t j = 0; j < arenas_per_call; ++j)
                    if (memcmp(got.data() + (size_t)j * G, expected + (size_t)qi * G, (size_t)G * 8) != 0) { mismatches++; break; }
            // (a thread's own slice of the latency buffer: a shared cursor would be one more cache line all threads write)
            if (mine < per_thread) lat_ns[(uint64_t)tid * per_thread + mine] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            ++mine;
            qi = (qi + 1) % n_queries;
            ai = (ai + arenas_per_call) % n_arena_ids;
        }
        calls += mine;
        recorded[tid] = std::min<uint64_t>(mine, per_thread);
        if (scope != ctx) bsg_close(scope);
    };
    for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(body, t);
    while (ready.load() < n_threads) std::this_thread::yield();
    auto cpu_now = []() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
    const double c0 = cpu_now();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(1, std::memory_order_release);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    go.store(2, std::memory_order_release);
    for (auto &t : th) t.join();
    const auto t1 = std::chrono::steady_clock::now();
    out->cpu_seconds = cpu_now() - c0;
    out->calls = calls.load();
    out->mismatches = mismatches.load();
    out->errors = errors.load();
    out->seconds = std::chrono::duration<double>(t1 - t0).count();
    uint64_t w = 0;                                   // pack the threads' slices to the front
    for (uint32_t t = 0; t < n_threads; ++t)
        for (uint64_t i = 0; i < recorded[t]; ++i) lat_ns[w++] = lat_ns[(uint64_t)t * per_thread + i];
    out->n_lat = w;
    return 0;
}

}  // extern "C"
