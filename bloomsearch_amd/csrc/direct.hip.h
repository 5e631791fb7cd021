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

__host__ __device__ inline uint32_t direct_lds_bytes(uint32_t Wt, uint32_t max_depth)
{
    return (Wt * 64u + max_depth * (uint32_t)kEvalThreads) * 8u;
}

// The shared body: th = SoA term hashes th[j * a.Tp + pos]; prog = the chunk's programs, op j of lane l at prog[j * prog_stride + l];
// len = ops of the longest program.  A kind's terms sit at positions term_begin .. term_begin + term_count of th AND of VT.
// g = the workgroup's 64-block group of the arena; n_wg = workgroups of the launch (the doorbell rings when all are done).
__device__ __forceinline__ void direct_body(const DirectArgs &a, const uint64_t *th, const uint32_t *prog, uint32_t prog_stride, uint32_t len,
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
    for (uint32_t y = 0; y < a.n_kinds; ++y) {
        DevDesc d{0, 0, 0, 0, 0};
        if (valid) d = ar.desc[(uint64_t)b * 3 + a.kind[y]];
        const uint64_t *src = ar.words + d.word_off;
        const uint32_t t0 = a.term_begin[y];
        for (uint32_t t = wave; t < a.term_count[y]; t += n_waves) {       // wave-uniform term: its hashes are scalar loads
            const uint64_t h0 = th[t0 + t], h1 = th[(uint64_t)a.Tp + t0 + t];
            const uint64_t h2 = th[2ull * a.Tp + t0 + t], h3 = th[3ull * a.Tp + t0 + t];
            // a nil filter cannot disqualify (query_exec.go:137-151); rows past the arena's end are masked at the end
            bool pass = true;
            uint32_t i = 0;
            const uint32_t k = d.m ? d.k : 0u;
            uint64_t s2 = 0, s3 = 0;                                      // i * h2, i * h3 as running sums
#ifndef BSG_DIRECT_TRIP
#define BSG_DIRECT_TRIP 12
#endif
            constexpr uint32_t kTrip = BSG_DIRECT_TRIP;                  // locations whose word reads are in flight together (lab: -DBSG_DIRECT_TRIP)
            while (__ballot(pass && i < k) != 0ull) {
                uint64_t w[kTrip]; uint32_t bit[kTrip]; bool live[kTrip];
#pragma unroll
                for (uint32_t u = 0; u < kTrip; ++u) {
                    const uint32_t r = (i + u) & 3u;
                    const uint64_t x = (((i + u) & 1u) ? h1 : h0) + ((r == 1u || r == 2u) ? s3 : s2);
                    s2 += h2; s3 += h3;
                    live[u] = pass && i + u < k;
                    const uint64_t loc = live[u] ? mod_m(x, d.m, d.magic) : 0;
                    bit[u] = (uint32_t)loc & 63u;
                    w[u] = live[u] ? src[loc >> 6] : 0;
                }
#pragma unroll
                for (uint32_t u = 0; u < kTrip; ++u)
                    if (live[u] && !((w[u] >> bit[u]) & 1ull)) pass = false;
                i += kTrip;
            }
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
    direct_body(a, a.th, a.prog, kEvalThreads, a.chunk_len[0], t.ar[blockIdx.z], lds64, blockIdx.x, gridDim.x * gridDim.z);
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
    direct_body(q.a, th, prog, q.stride, q.len, *ar, lds64, blockIdx.x, gridDim.x * gridDim.z);
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
    uint32_t n_jobs, n_wg;
    uint32_t *done_count; uint64_t *flag; uint64_t seq;
};

__global__ __launch_bounds__(kEvalThreads) void k_query_jobs(const JobsArgs j)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    // (a binary search over the records' first workgroups cost every workgroup ~7 dependent loads in front of its first useful one:
    // 81 jobs 84 us; the table costs 4 bytes per workgroup)
    const QJob &J = j.jobs[j.wg_job[blockIdx.x]];
    DirectArgs a{};
    a.out = j.out + J.out_off;
    a.Tp = J.th_stride; a.Wt = 1; a.n_queries = J.n_queries; a.Lmax = J.len; a.max_depth = J.max_depth; a.n_kinds = J.n_kinds;
#pragma unroll
    for (uint32_t y = 0; y < 3; ++y) { a.kind[y] = J.kind[y]; a.term_begin[y] = J.term_begin[y]; a.term_count[y] = J.term_count[y]; }
    a.n_arenas = 1;
    a.done_count = j.done_count; a.flag = j.flag; a.seq = j.seq;
    const ArenaRef ar{J.words, J.desc, J.n_blocks, 0u};
    direct_body(a, j.tab + J.th_off, reinterpret_cast<const uint32_t *>(j.tab) + J.prog_off, J.n_queries, J.len, ar, lds64, blockIdx.x - J.wg0, j.n_wg);
}

// The completion doorbell of a combiner cycle: ONE system-scope store behind the cycle's last dispatch (same stream, in order) into
// page-locked memory the collector polls — no runtime call while waiting (an event polled with hipEventQuery takes the runtime's
// locks the next collector's enqueue needs), and one fence per cycle instead of one per workgroup.
__global__ void k_ring(uint64_t *flag, uint64_t seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace bsg
