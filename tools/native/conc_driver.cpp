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
#include <algorithm>
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

// ---- the resident file-arena cache under concurrent callers (cache_api.inc) ----
// T threads, each looping: pick a file, a candidate block set of it (all blocks, or a run, or a stride: different callers ask for
// different subsets, which is what widens a resident arena) and a query; bsg_file_arena_acquire; on a miss load the union of the
// resident arena's blocks and the candidates from the file's stored section bytes (bsg_arena_load_sections) and publish it; probe
// the leased arena with bsg_query; compare every candidate's verdict with the expected bit; release.  Every `forget_every`-th call
// of a thread tombstones the file it just used (bsg_file_arena_forget) while other threads may hold leases on it.
struct cache_file {
    const uint8_t *region;      // the file's block-filter sections, back to back
    const uint64_t *sec_off;    // [n_blocks + 1] offsets into region
    uint32_t n_blocks;
};
struct cache_result {
    uint64_t calls, mismatches, errors, hits, misses, forgets;
    double seconds;
};

__attribute__((visibility("default")))
int32_t cache_run(bsg_ctx *ctx, uint32_t n_threads, double seconds, const conc_query *queries, uint32_t n_queries, const cache_file *files,
                  uint32_t n_files, const uint64_t *const *expected /* [n_files] -> [n_queries][G_f] */, uint32_t forget_every, uint64_t seed,
                  cache_result *out)
{
    std::atomic<uint64_t> calls{0}, mismatches{0}, errors{0}, hits{0}, misses{0}, forgets{0};
    std::atomic<int> go{0};
    std::atomic<uint32_t> ready{0};
    auto key_of = [](uint32_t b) { return (uint64_t)b * 4096u + 17u; };            // block b's key (RowDataOffset): strictly ascending in b
    auto body = [&](uint32_t tid) {
        bsg_ctx *scope = nullptr;
        if (bsg_scope_open(ctx, &scope) != BSG_OK) { errors++; scope = ctx; }
        uint64_t rng = seed * 0x9E3779B97F4A7C15ull + tid * 0xD1B54A32D192ED03ull + 1;
        auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        std::vector<uint64_t> keys, have_k, have_b, have_e, all_k, all_b, all_e, sec_off, got;
        std::vector<uint32_t> rows, cand;
        std::vector<int32_t> status;
        std::vector<uint8_t> region;
        ready++;
        while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        uint64_t mine = 0;
        while (go.load(std::memory_order_relaxed) == 1) {
            const uint32_t f = (uint32_t)(next() % n_files), qi = (uint32_t)(next() % n_queries);
            const cache_file &F = files[f];
            const uint8_t fkey[4] = {(uint8_t)f, (uint8_t)(f >> 8), (uint8_t)(f >> 16), (uint8_t)(f >> 24)};
            // the candidate blocks: everything (half the calls), a run, or every third block
            cand.clear();
            const uint32_t shape = (uint32_t)(next() % 4);
            if (shape < 2) for (uint32_t b = 0; b < F.n_blocks; ++b) cand.push_back(b);
            else if (shape == 2) { const uint32_t lo = (uint32_t)(next() % F.n_blocks), n = 1 + (uint32_t)(next() % (F.n_blocks - lo)); for (uint32_t b = lo; b < lo + n; ++b) cand.push_back(b); }
            else for (uint32_t b = (uint32_t)(next() % 3); b < F.n_blocks; b += 3) cand.push_back(b);
            if (cand.empty()) cand.push_back(0);
            keys.resize(cand.size());
            for (size_t i = 0; i < cand.size(); ++i) keys[i] = key_of(cand[i]);
            rows.assign(cand.size(), 0);
            uint64_t lease = 0, arena = 0;
            bool ok = bsg_file_arena_acquire(scope, fkey, 4, keys.data(), (uint32_t)keys.size(), &lease, &arena, nullptr, rows.data()) == BSG_OK;
            if (ok && lease) {
                hits++;
            } else if (ok) {
                misses++;
                // union of what is resident and the candidates
                uint32_t n = 0;
                have_k.clear(); have_b.clear(); have_e.clear();
                if (bsg_file_arena_have(scope, fkey, 4, nullptr, nullptr, nullptr, 0, &n) == BSG_OK && n) {
                    have_k.resize(n); have_b.resize(n); have_e.resize(n);
                    uint32_t n2 = 0;
                    if (bsg_file_arena_have(scope, fkey, 4, have_k.data(), have_b.data(), have_e.data(), n, &n2) != BSG_OK || n2 != n) have_k.clear();   // changed in between: load the candidates alone
                }
                all_k.clear();
                size_t i = 0, j = 0;
                while (i < keys.size() || j < have_k.size()) {
                    if (j == have_k.size() || (i < keys.size() && keys[i] <= have_k[j])) { if (j < have_k.size() && keys[i] == have_k[j]) ++j; all_k.push_back(keys[i++]); }
                    else all_k.push_back(have_k[j++]);
                }
                region.clear(); sec_off.assign(1, 0); all_b.clear(); all_e.clear();
                for (uint64_t k : all_k) {
                    const uint32_t b = (uint32_t)((k - 17u) / 4096u);
                    region.insert(region.end(), F.region + F.sec_off[b], F.region + F.sec_off[b + 1]);
                    sec_off.push_back(region.size());
                    all_b.push_back(F.sec_off[b]); all_e.push_back(F.sec_off[b + 1]);
                }
                status.assign(all_k.size(), 0);
                ok = bsg_arena_load_sections(scope, region.data(), region.size(), sec_off.data(), (uint32_t)all_k.size(), status.data(), &arena) == BSG_OK;
                if (ok) {
                    ok = bsg_file_arena_publish(scope, fkey, 4, arena, all_k.data(), all_b.data(), all_e.data(), status.data(), (uint32_t)all_k.size(), &lease, nullptr) == BSG_OK;
                    if (!ok) bsg_arena_free(scope, arena);
                }
                if (ok) {
                    for (size_t c = 0; c < keys.size(); ++c) rows[c] = (uint32_t)(std::lower_bound(all_k.begin(), all_k.end(), keys[c]) - all_k.begin());
                }
            }
            if (!ok) { errors++; if (lease) bsg_file_arena_release(scope, lease); ++mine; continue; }
            // probe the leased arena: its survivors row has ceil(arena blocks / 64) words, at most the file's
            got.assign((F.n_blocks + 63) / 64, 0);
            const conc_query &q = queries[qi];
            uint32_t off2[2] = {0, q.n_ops};
            if (bsg_query(scope, &arena, 1, q.term_bytes, q.term_off, q.term_kinds, q.n_terms, q.prog_ops, off2, 1, got.data()) != BSG_OK) errors++;
            else {
                const uint32_t Gf = (F.n_blocks + 63) / 64;
                const uint64_t *exp = expected[f] + (size_t)qi * Gf;
                for (size_t c = 0; c < cand.size(); ++c) {
                    const uint32_t r = rows[c], b = cand[c];
                    if (((got[r >> 6] >> (r & 63)) & 1) != ((exp[b >> 6] >> (b & 63)) & 1)) { mismatches++; break; }
                }
            }
            if (forget_every && mine % forget_every == forget_every - 1) { bsg_file_arena_forget(scope, fkey, 4); forgets++; }   // ... while we (and maybe others) still hold a lease
            if (bsg_file_arena_release(scope, lease) != BSG_OK) errors++;
            ++mine;
        }
        calls += mine;
        if (scope != ctx) bsg_close(scope);
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(body, t);
    while (ready.load() < n_threads) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(1, std::memory_order_release);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    go.store(2, std::memory_order_release);
    for (auto &t : th) t.join();
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out->calls = calls.load(); out->mismatches = mismatches.load(); out->errors = errors.load();
    out->hits = hits.load(); out->misses = misses.load(); out->forgets = forgets.load();
    return 0;
}

}  // extern "C"
