// direct.hip.h — one interactive query (or a handful) in ONE dispatch.
// evaluateBlockFilters for a single Query() call tests a few terms per block (query_exec.go:128-158, 572-615): 30 bit
// tests per block for a 3-term query.  Through k_probe_terms + k_eval_programs + the copy of the survivors that is three
// enqueues of ~5 us launch latency each around ~10 us of kernels (measured 30-34 us per synchronous query, bench `q1`).
// Here a workgroup owns 64 consecutive blocks of one arena end to end: lane = block, the waves share the terms; every
// (block, term) is tested with <= k word reads straight from L2 / HBM (the gather regime of SURVEY 8d: a 35 KB bitset is
// never streamed for 30 bits), __ballot turns a term's 64 verdicts into the 64-block mask the program interpreter works
// on, and the same workgroup runs the batch's programs and writes the survivor words — into device memory or straight
// into page-locked host memory, so a query costs one launch and one wait.
// Taken for batches of <= 256 queries with few distinct terms (bsg_set_lab key 3); everything else keeps the streaming kernels.
#pragma once
#include "kernels.hip.h"

namespace bsg {

#ifndef BSG_DIRECT_TRIP
#define BSG_DIRECT_TRIP 12
#endif

struct DirectArgs {
    const uint64_t *th;          // SoA term hashes: th[j * Tp + t]
    const uint32_t *prog;        // the batch's single 256-query chunk, lane-interleaved: prog[j * 256 + lane]
    const uint32_t *chunk_len;
    uint64_t *out;               // arena i: [n_queries][G_i] at out + ar[i].out_off (device memory or mapped host memory)
    uint32_t Tp, Wt, n_queries, Lmax, max_depth, n_kinds;
    uint32_t kind[3], term_begin[3], term_count[3];
    uint32_t n_arenas;
    // completion doorbell (host output only): the last workgroup to finish writes `seq` into page-locked host memory, so the
    // host learns of the result by reading its own memory instead of asking the runtime (hipStreamSynchronize: ~5 us)
    uint32_t *done_count;        // device word, 0 between launches
    uint64_t *flag;              // host-mapped; nullptr: no doorbell
    uint64_t seq;
};

// The kinds a dispatch probes, as the body wants them: three (kind, first position, terms) triples packed into words — picked apart
// with shifts, so that a body working from a job record keeps them in registers (three-element arrays indexed by a loop variable
// put the whole argument struct into scratch memory: 128 bytes per lane in k_query_jobs).
struct KindsPk {
    uint32_t kind;               // kind of triple y in byte y
    uint64_t begin, count;       // 16 bits per triple
};
__device__ __forceinline__ KindsPk pack_kinds(const uint32_t *kind, const uint32_t *begin, const uint32_t *count, uint32_t n_kinds)
{
    KindsPk k{0, 0, 0};
#pragma unroll
    for (uint32_t y = 0; y < 3; ++y)
        if (y < n_kinds) { k.kind |= kind[y] << (8 * y); k.begin |= (uint64_t)begin[y] << (16 * y); k.count |= (uint64_t)count[y] << (16 * y); }
    return k;
}

__host__ __device__ inline uint32_t direct_lds_bytes(uint32_t Wt, uint32_t max_depth)
{
    return (Wt * 64u + max_depth * (uint32_t)kEvalThreads) * 8u;
}

// The shared body: th = SoA term hashes th[j * a.Tp + pos]; prog = the chunk's programs, op j of lane l at prog[j * prog_stride + l];
// len = ops of the longest program.  A kind's terms sit at positions term_begin .. term_begin + term_count of th AND of VT.
// g = the workgroup's 64-block group of the arena; n_wg = workgroups of the launch (the doorbell rings when all are done).
template <uint32_t FIRST = 0, uint32_t TRIP = BSG_DIRECT_TRIP>
__device__ __forceinline__ void direct_body(const DirectArgs &a, const KindsPk K, const uint64_t *th, const uint32_t *prog, uint32_t prog_stride, uint32_t len,
                                            const ArenaRef &ar, uint64_t *lds64, const uint32_t g, const uint32_t n_wg)
{
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (g < ar.G()) {                                              // (arenas of a group may differ in size)
    constexpr uint32_t n_waves = kEvalThreads / kWave;
    uint64_t *VT = lds64;                                        // VT[position]: 64-block mask of one term
    uint64_t *stk = lds64 + (uint64_t)a.Wt * 64 + tid;           // per-lane stack, stride kEvalThreads
    for (uint32_t i = tid; i < a.Wt * 64; i += kEvalThreads) VT[i] = 0;
    // the program words do not depend on the verdicts: request them before the bit tests
    constexpr uint32_t kPre = 8;
    uint32_t pre[kPre];
    const bool has_prog = tid < prog_stride;                     // (the compact layout of k_query_direct holds n_queries lanes only)
    const uint32_t *P = prog + tid;
#pragma unroll
    for (uint32_t j = 0; j < kPre; ++j) pre[j] = (has_prog && j < a.Lmax) ? P[(uint64_t)j * prog_stride] : (7u << 28);
    __syncthreads();

    const uint32_t b = g * 64 + lane;
    const bool valid = b < ar.n_blocks;
    // the terms of all kinds dealt round-robin to the waves: a 3-term query with one term per kind runs its three
    // descriptor -> words chains side by side on three waves instead of one after the other on the first
    const uint32_t c0 = (uint32_t)K.count & 0xFFFFu, c1 = (uint32_t)(K.count >> 16) & 0xFFFFu, c2 = (uint32_t)(K.count >> 32) & 0xFFFFu;
    const uint32_t total = c0 + c1 + c2;
    uint32_t cur_y = ~0u;
    DevDesc d{0, 0, 0, 0, 0};
    const uint64_t *src = ar.words;
    {
        for (uint32_t q = wave; q < total; q += n_waves) {               // wave-uniform term: its hashes are scalar loads
            uint32_t y = 0, t = q;
            if (t >= c0) { t -= c0; y = 1; if (t >= c1) { t -= c1; y = 2; } }
            if (y != cur_y) {
                cur_y = y;
                const uint32_t kind = (K.kind >> (8 * y)) & 0xFFu;
                d = DevDesc{0, 0, 0, 0, 0};
                if (valid) d = ar.desc[(uint64_t)b * 3 + kind];
                src = ar.words + d.word_off;
            }
            const uint32_t t0 = (uint32_t)(K.begin >> (16 * y)) & 0xFFFFu;
            const uint64_t h0 = th[t0 + t], h1 = th[(uint64_t)a.Tp + t0 + t];
            const uint64_t h2 = th[2ull * a.Tp + t0 + t], h3 = th[3ull * a.Tp + t0 + t];
            // a nil filter cannot disqualify (query_exec.go:137-151); rows past the arena's end are masked at the end
            bool pass = true;
            uint32_t i = 0;
            const uint32_t k = d.m ? d.k : 0u;
            uint64_t s2 = 0, s3 = 0;                                      // i * h2, i * h3 as running sums
            constexpr uint32_t kTrip = TRIP;                             // locations whose word reads are in flight together (lab: -DBSG_DIRECT_TRIP)
            auto trip = [&](auto n_c) {
                constexpr uint32_t N = decltype(n_c)::value;
                uint64_t w[N]; uint32_t bit[N]; bool live[N];
#pragma unroll
                for (uint32_t u = 0; u < N; ++u) {
                    const uint32_t r = (i + u) & 3u;
                    const uint64_t x = (((i + u) & 1u) ? h1 : h0) + ((r == 1u || r == 2u) ? s3 : s2);
                    s2 += h2; s3 += h3;
                    live[u] = pass && i + u < k;
                    const uint64_t loc = live[u] ? mod_m(x, d.m, d.magic) : 0;
                    bit[u] = (uint32_t)loc & 63u;
                    w[u] = live[u] ? src[loc >> 6] : 0;
                }
#pragma unroll
                for (uint32_t u = 0; u < N; ++u)
                    if (live[u] && !((w[u] >> bit[u]) & 1ull)) pass = false;
                i += N;
            };
            // FIRST > 0: a short first trip — an absent term fails within a few bits (half of a filter's bits are set) and only
            // the lanes still passing read on: fewer line fetches for more dependent rounds.  A long job list is bound by the
            // fetches themselves (below); a lone call (16 workgroups) by its rounds, and keeps the single trip.
            if constexpr (FIRST > 0) trip(std::integral_constant<uint32_t, FIRST>{});
            while (__ballot(pass && i < k) != 0ull) trip(std::integral_constant<uint32_t, kTrip>{});
            const uint64_t mask = __ballot(pass);
            if (lane == 0) VT[t0 + t] = mask;                            // (a batch's kind segments start on word boundaries: t0 is their slot * 64)
        }
    }
    __syncthreads();

    // evalBloomExpression over 64-block masks (the interpreter of k_eval_programs: 0 TERM | 1 AND2 | 2 OR2 | 3 TRUE | 4 FALSE | 7 NOP)
    uint64_t top = ~0ULL;                                                // empty program == nil query == true
    uint32_t sp = 0;
    auto step = [&](uint32_t op) {
        const uint32_t opc = op >> 28;
        if (opc == 7u) return;
        if (opc == 1u || opc == 2u) {
            --sp;
            const uint64_t under = stk[(uint64_t)(sp - 1) * kEvalThreads];
            top = (opc == 1u) ? (under & top) : (under | top);
        } else {
            if (sp > 0) stk[(uint64_t)(sp - 1) * kEvalThreads] = top;
            ++sp;
            top = (opc == 0u) ? VT[op & 0x0FFFFFFFu] : (opc == 3u ? ~0ULL : 0ULL);
        }
    };
    if (has_prog) {
#pragma unroll
        for (uint32_t j = 0; j < kPre; ++j) if (j < len) step(pre[j]);
        for (uint32_t j = kPre; j < len; ++j) step(P[(uint64_t)j * prog_stride]);
    }
    const uint32_t nvalid = ar.n_blocks - g * 64;
    if (tid < a.n_queries)
        a.out[ar.out_off(a.n_queries) + (uint64_t)tid * ar.G() + g] = top & (nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1));
    }
    if (a.flag) {
        __threadfence_system();                                  // this workgroup's survivor words reach the host before the count moves
        __syncthreads();
        if (tid == 0) {
            if (__hip_atomic_fetch_add(a.done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_wg) {
                __hip_atomic_store(a.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// grid = (most 64-block groups of any arena, 1, arenas); 256 threads
__global__ __launch_bounds__(kEvalThreads) void k_probe_direct(const DirectArgs a, const ArenaTable<kMaxGroupArenas> t)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    direct_body(a, pack_kinds(a.kind, a.term_begin, a.term_count, a.n_kinds), a.th, a.prog, kEvalThreads, a.chunk_len[0], t.ar[blockIdx.z], lds64, blockIdx.x, gridDim.x * gridDim.z);
}

// ---- the same dispatch with NOTHING uploaded beforehand: bsg_query ----
// One Query() of the reference hashes its few terms inside TestString and walks its expression per block
// (query_exec.go:128-158).  Through the batch API that is a device launch to hash three strings, five small uploads for
// the batch object, and the probe.  Here the term hashes (computed on the host by the same base_hashes) and the lowered
// programs ride IN THE KERNEL ARGUMENTS: strings in, survivors out, one dispatch, nothing allocated, nothing uploaded.
constexpr uint32_t kQueryMaxTerms = 16;        // distinct terms of one call (th: 4 x 16 x 8 B of arguments)
constexpr uint32_t kQueryMaxProgWords = 128;   // longest program x queries of one call
constexpr uint32_t kQueryMaxArenas = 32;       // arenas (per device) one call covers; more take the batch path

struct QueryKernArgs {
    DirectArgs a;                              // th / prog / chunk_len unused: the kernel reads the arrays below
    uint32_t len, stride, pad0, pad1;          // ops of the longest program; lanes per op row (= n_queries)
    uint64_t th[4 * kQueryMaxTerms];           // th[j * kQueryMaxTerms + pos]
    uint32_t prog[kQueryMaxProgWords];         // prog[j * stride + q]
    ArenaTable<kQueryMaxArenas> t;
};
static_assert(sizeof(QueryKernArgs) <= 4096, "kernel arguments must stay within 4 KB");

__global__ __launch_bounds__(kEvalThreads) void k_query_direct(const QueryKernArgs q)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    // the arrays are indexed dynamically: read them where they already are — the kernel-argument segment — instead of
    // letting the compiler copy a by-value struct into scratch
    const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();      // (constant address space -> generic)
    const uint64_t *th = reinterpret_cast<const uint64_t *>(ka + offsetof(QueryKernArgs, th));
    const uint32_t *prog = reinterpret_cast<const uint32_t *>(ka + offsetof(QueryKernArgs, prog));
    const ArenaRef *ar = reinterpret_cast<const ArenaRef *>(ka + offsetof(QueryKernArgs, t)) + blockIdx.z;
    direct_body(q.a, pack_kinds(q.a.kind, q.a.term_begin, q.a.term_count, q.a.n_kinds), th, prog, q.stride, q.len, *ar, lds64, blockIdx.x, gridDim.x * gridDim.z);
}

// ---- MANY such calls in one dispatch: the job list of a combiner cycle (combine_api.inc) ----
// Concurrent bsg_query callers (the reference's file workers: one goroutine per candidate file, several Query() calls at once,
// query_exec.go:303-357, 427-431) each ask for one small query set against a few arenas.  A JOB is one (call, arena) pair; a
// workgroup is one 64-block group of one job and runs the body above unchanged — its term hashes and programs lie in a table
// uploaded once per cycle (a call's jobs share them), its rows go to the job's own place in page-locked memory, and ONE doorbell
// rings for the whole list.  Cost follows the pairs asked for, not the product of all queries and all arenas of the cycle.
struct QJob {
    const uint64_t *words;       // the arena shard
    const DevDesc *desc;
    uint32_t n_blocks;
    uint32_t wg0;                // first workgroup of the job (jobs ascending; a job owns ceil(n_blocks / 64) workgroups)
    uint64_t out_off;            // first word of the job's rows [n_queries][G] in the result
    uint32_t th_off;             // u64 index into the table: th[j * th_stride + pos], j < 4
    uint32_t prog_off;           // u32 index into the table (as u32): prog[j * n_queries + q]
    uint16_t n_queries, len, max_depth, n_kinds;
    uint8_t kind[4], term_begin[4], term_count[4];
    uint32_t th_stride;          // the call's terms (its hashes lie packed: 4 rows of th_stride words)
};
static_assert(sizeof(QJob) == 64, "one job record per 64-byte line");

struct JobsArgs {
    const QJob *jobs;            // device memory (the cycle's table; hashes and programs behind the records)
    const uint64_t *tab;         // the same table, as u64 / u32 words
    const uint32_t *wg_job;      // the job of every workgroup (in the same table): one load instead of a search over the records
    uint64_t *out;               // page-locked host memory
    uint32_t *done_count; uint64_t *flag; uint64_t seq;
    uint32_t n_jobs, n_wg;
    uint32_t wg_at, pad;         // k_query_jobs_inline: byte offset of wg_job in the table (the three pointers above are unused there)
};

// What bounds a long list (PMC, 5 000-7 000 workgroups per dispatch: tools/conc_pmc.sh): every bit test is an L2 miss (TCC_MISS 1 223 per
// workgroup against 375 hits, 64 bytes each by FETCH_SIZE, no TLB misses), the waves wait all their life (SQ_WAIT_ANY / SQ_WAVE_CYCLES
// 0.7-1.0, instructions issue in 3-7 % of it), and a workgroup costs 23.5 ns whatever the occupancy or the number of dependent rounds:
// 52 x 10^9 scattered sector reads per second — the HBM's rate for scattered accesses (each opens a DRAM row), 3.3 TB/s of sectors for
// 0.42 TB/s of useful words.  Hence bits are tested two at a time (FIRST = TRIP = 2: 2.7 fetches per absent term instead of 10; measured
// against 4 + 6: +3 ... 8 % on lists of 850-10 000 workgroups where one term in three is present in every block, level on short lists),
// 98 -> ~40 registers, and nothing in scratch memory (the kinds' triples packed into words above: as arrays they cost 128 bytes per
// lane and 15 % of the kernel's time).
#ifndef BSG_JOBS_FIRST
#define BSG_JOBS_FIRST 2
#endif
#ifndef BSG_JOBS_TRIP
#define BSG_JOBS_TRIP 2
#endif
__device__ __forceinline__ void jobs_body(const JobsArgs &j, const QJob *jobs, const uint64_t *tab, const uint32_t *wg_job, uint64_t *lds64)
{
    // (a binary search over the records' first workgroups cost every workgroup ~7 dependent loads in front of its first useful one:
    // 81 jobs 84 us; the table costs 4 bytes per workgroup)
    const QJob &J = jobs[wg_job[blockIdx.x]];
    DirectArgs a{};
    a.out = j.out + J.out_off;
    a.Tp = J.th_stride; a.Wt = 1; a.n_queries = J.n_queries; a.Lmax = J.len; a.max_depth = J.max_depth; a.n_kinds = J.n_kinds;
    KindsPk K{0, 0, 0};
#pragma unroll
    for (uint32_t y = 0; y < 3; ++y)
        if (y < J.n_kinds) { K.kind |= (uint32_t)J.kind[y] << (8 * y); K.begin |= (uint64_t)J.term_begin[y] << (16 * y); K.count |= (uint64_t)J.term_count[y] << (16 * y); }
    a.n_arenas = 1;
    a.done_count = j.done_count; a.flag = j.flag; a.seq = j.seq;
    const ArenaRef ar{J.words, J.desc, J.n_blocks, 0u};
    direct_body<BSG_JOBS_FIRST, BSG_JOBS_TRIP>(a, K, tab + J.th_off, reinterpret_cast<const uint32_t *>(tab) + J.prog_off, J.n_queries, J.len, ar, lds64, blockIdx.x - J.wg0, j.n_wg);
}

__global__ __launch_bounds__(kEvalThreads) void k_query_jobs(const JobsArgs j)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    jobs_body(j, j.jobs, j.tab, j.wg_job, lds64);
}

// A SHORT job list (a cycle of a handful of callers: the table of ~16 three-term single-arena calls) rides in the kernel arguments
// like k_query_direct's: no table upload in front of the dispatch — one enqueue less per cycle, and the copy's ~3 us on the stream.
constexpr uint32_t kJobsInlineBytes = 4096 - (uint32_t)sizeof(JobsArgs);
struct JobsInlineArgs {
    JobsArgs a;
    uint8_t table[kJobsInlineBytes];           // [jobs][per call: th | prog][job of every workgroup]
};
static_assert(sizeof(JobsInlineArgs) == 4096 && offsetof(JobsInlineArgs, table) % 8 == 0, "kernel arguments must stay within 4 KB");

__global__ __launch_bounds__(kEvalThreads) void k_query_jobs_inline(const JobsInlineArgs q)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const char *tb = (const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(JobsInlineArgs, table);
    jobs_body(q.a, reinterpret_cast<const QJob *>(tb), reinterpret_cast<const uint64_t *>(tb), reinterpret_cast<const uint32_t *>(tb + q.a.wg_at), lds64);
}

// The completion doorbell of a combiner cycle: ONE system-scope store behind the cycle's last dispatch (same stream, in order) into
// page-locked memory the collector polls — no runtime call while waiting (an event polled with hipEventQuery takes the runtime's
// locks the next collector's enqueue needs), and one fence per cycle instead of one per workgroup.
__global__ void k_ring(uint64_t *flag, uint64_t seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace bsg
