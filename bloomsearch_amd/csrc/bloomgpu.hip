// bloomgpu.hip — C-ABI implementation of include/bloomgpu.h (gfx950 only).
//
// Host-side plumbing around the kernels in kernels.hip.h: contexts, per-device
// streams, filter arenas (sharded round-robin over the context's devices),
// compiled query batches, probe / build / OR-reduce entry points.
// There is no CPU fallback anywhere in this file: without a GPU every compute
// entry point fails with BSG_E_NODEVICE / BSG_E_HIP.
#include "bloomgpu.h"
#include "bloomgpu_lab.h"
#include "kernels.hip.h"
#include "ingest.hip.h"
#include "bin_build.hip.h"
#include "direct.hip.h"
#include "match.hip.h"
#include "host/combiner_sync.hpp"
#include "host/text.hpp"   // the host walker's Unicode tables: the device defers to the same data
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

using bsg::DevDesc;

// Error reporting that survives a caller migrating between OS threads (a goroutine between two cgo calls):
// every failure is recorded (1) in the slot of the scope / context the failing call was made on — guarded by a
// mutex, read back with bsg_last_error(scope) or bsg_last_error_copy from ANY thread — and (2) in a thread-local
// string that only serves calls which have no context yet (bsg_open, bsg_estimate_parameters, bsg_sections_size).
thread_local std::string g_err;
thread_local std::string g_ret;          // what bsg_last_error hands out: a private copy, valid until this thread's next call
thread_local bsg_ctx *tl_scope = nullptr;
void record_error(bsg_ctx *scope, const char *msg);   // defined after bsg_ctx

int32_t fail(int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (tl_scope) record_error(tl_scope, buf);
    return code;
}

// First line of every entry point that takes a context: remembers the scope the call was made on for fail(),
// rejects NULL, and rebinds `ctx` to the root context a scope aliases.  Nested API calls keep the outermost scope.
struct ScopeGuard {
    bool owner;
    explicit ScopeGuard(bsg_ctx *s) : owner(tl_scope == nullptr && s != nullptr) { if (owner) tl_scope = s; }
    ~ScopeGuard() { if (owner) tl_scope = nullptr; }
};
bsg_ctx *root_of(bsg_ctx *c);
#define BSG_ENTER(ctx)                                            \
    ScopeGuard scope_guard_(ctx);                                 \
    if (!(ctx)) return fail(BSG_E_INVALID, "ctx is null");        \
    (ctx) = root_of(ctx)

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(BSG_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

constexpr uint32_t kLdsBudget = 144 * 1024;   // dynamic LDS a workgroup may request (of 160 KiB per CU; opt-in above 64 KiB)
constexpr uint32_t kLdsCapWords = kLdsBudget / 8;  // staged-filter cap before the per-launch head is taken off
constexpr uint64_t kAlignWords = 16;          // filters start on 128-byte boundaries in HBM
constexpr uint32_t kBuildSliceEntries = 8192; // entries per workgroup for non-staged builds
constexpr uint64_t kBinScratchBytes = 16ull << 30;   // locations (4 bytes each) one binned build may park in HBM
constexpr uint64_t kMaxHashCount = 1024;      // k above this is rejected (EstimateParameters: 30 at p = 1e-9, 100 at 1e-30)

uint64_t barrett_magic(uint64_t m)
{
    if (m <= 1) return ~0ULL;
    // floor(2^64 / m) without 128-bit division: (2^64 - 1) / m, +1 iff m divides 2^64 (m power of two)
    uint64_t q = ~0ULL / m;
    if ((m & (m - 1)) == 0) q += 1;
    return q;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = std::max<size_t>(n, cap + cap / 2);   // growing scratch: do not reallocate on every slightly larger call
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// start/stop timestamps of the two dispatches themselves (hipExtLaunchKernel), i.e. the same
// begin/end a rocprofv3 kernel trace reports — not events bracketing the launches.
struct EventTriple { hipEvent_t k1s, k1e, k2s, k2e; uint64_t bytes; uint32_t n_arenas; bool has_k1, has_k2, fused, folded; };

// Short-lived device buffers of the ingest / match / encode calls come from a per-device cache instead of
// hipMalloc / hipFree (a flush of 1 000 rows made ~16 of each: 0.8 ms of fixed cost, 2/3 of the call).  Blocks are kept
// by size class; everything is guarded by Device::mu, which those paths hold anyway.
struct DevPool {
    std::multimap<size_t, void *> idle;            // size class -> block
    std::map<void *, size_t> live;                 // block -> size class
    size_t idle_bytes = 0;
    static constexpr size_t kMaxIdleBytes = 48ull << 30;   // of 288 GB: a 10 M-row flush holds ~10 GB of tables + rows, and hipMalloc of that costs 0.2-0.5 s
    static size_t size_class(size_t n)
    {
        n = std::max<size_t>(n, 256);
        if (n <= (64ull << 20)) { size_t c = 256; while (c < n) c <<= 1; return c; }
        return (n + (64ull << 20) - 1) / (64ull << 20) * (64ull << 20);
    }
    hipError_t alloc(void **out, size_t n)
    {
        const size_t c = size_class(n);
        auto it = idle.find(c);
        if (it != idle.end()) {
            *out = it->second;
            idle.erase(it);
            idle_bytes -= c;
        } else {
            hipError_t e = hipMalloc(out, c);
            if (e != hipSuccess) {                   // make room and try once more
                trim(0);
                e = hipMalloc(out, c);
                if (e != hipSuccess) return e;
            }
        }
        live[*out] = c;
        return hipSuccess;
    }
    void free(void *p)
    {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) { (void)hipFree(p); return; }
        const size_t c = it->second;
        live.erase(it);
        if (idle_bytes + c > kMaxIdleBytes) { (void)hipFree(p); return; }
        idle.emplace(c, p);
        idle_bytes += c;
    }
    void trim(size_t keep_bytes)
    {
        while (idle_bytes > keep_bytes && !idle.empty()) {
            auto it = std::prev(idle.end());
            (void)hipFree(it->second);
            idle_bytes -= it->first;
            idle.erase(it);
        }
    }
};

// scratch of one combined query dispatch (combine_api.inc): the table's device block, its page-locked stage, the page-locked result
struct CmbSet { void *d_block = nullptr; size_t d_cap = 0; std::pair<uint64_t *, size_t> stage{nullptr, 0}, result{nullptr, 0}; hipEvent_t drained = nullptr; /* blocking-sync event of a wait that outlived its spin (finish_dispatches) */ };

struct Device {
    int id = 0;
    uint32_t n_cus = 256;
    DevPool pool;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;          // survivors leave here so the D2H of group i overlaps the kernels of group i+1
    hipStream_t aux_stream = nullptr;           // tail split: the evaluation of a run's first part beside the probe of its second
    hipEvent_t ev_aux[2] = {nullptr, nullptr};
    hipEvent_t ev_eval[2] = {nullptr, nullptr}; // out[slot] written (compute stream)
    hipEvent_t ev_copy[2] = {nullptr, nullptr}; // out[slot] copied out (copy stream)
    bool copy_busy[2] = {false, false};
    std::mutex mu;                 // serialises enqueue + scratch reuse on this device
    DevBuf<uint64_t> V[2];         // verdict scratch (two slots: bsg_probe_many software-pipelines launches)
    DevBuf<uint64_t> out[2];       // survivors scratch
    // arena records of dispatch groups beyond kMaxGroupArenas, in device memory (written by k_write_arena_table): the last few tables,
    // found again by their records (probe_api.inc: group_table); ext_of_slot: the table the group in scratch slot i uses
    struct TableEntry { uint64_t hash = 0; std::vector<bsg::ArenaRef> refs; DevBuf<bsg::ArenaRef> buf; uint64_t last_use = 0; };
    static constexpr size_t kTableCache = 64;
    std::vector<TableEntry> table_cache;
    uint64_t table_tick = 0;
    const bsg::ArenaRef *ext_of_slot[2] = {nullptr, nullptr};
    uint64_t fold_seq = 0;         // k_probe_eval: number of the last launch = the tag of its verdict entries
    DevBuf<uint8_t> stage_a;       // build/hash staging
    DevBuf<uint32_t> stage_off;
    DevBuf<uint64_t> stage_h;
    DevBuf<uint32_t> stage_fstart;
    DevBuf<DevDesc> stage_desc;
    DevBuf<bsg::BuildItem> stage_items;
    DevBuf<uint64_t> stage_words;
    DevBuf<uint8_t> stage_region;  // encoded filter sections
    std::vector<uint8_t *> idle_staging;   // pinned 4 MiB chunk buffers of finished arena streams
    // copy stream + events of finished arena streams (stream_api.inc): hipStreamCreate + hipStreamDestroy alone were ~3 ms of EVERY
    // bsg_arena_load_sections / arena stream, whatever its size — the miss path of the file-arena cache (tools/miss_lab.py, round 6)
    struct StreamKit { hipStream_t copy = nullptr; hipEvent_t copied = nullptr; hipEvent_t staged_free[2] = {nullptr, nullptr}; };
    std::vector<StreamKit> idle_kits;
    std::vector<std::pair<uint64_t *, size_t>> direct_bufs;   // idle page-locked result buffers of k_probe_direct (pointer, bytes)
    std::mutex cmb_mu;                        // guards cmb_sets only
    std::vector<CmbSet> cmb_sets;             // idle scratch sets of combined query dispatches
    uint32_t *d_direct_count = nullptr;       // finished-workgroup counter of k_probe_direct (0 between launches)
    uint64_t direct_seq = 0;                  // doorbell value of the last k_probe_direct launch
    std::vector<EventTriple> pending;
    std::vector<EventTriple> free_events;
    uint32_t *d_lower = nullptr;              // unicode.ToLower table for k_ingest_rows (512 KB)
    bsg::CrcConsts *d_crc = nullptr;          // CRC32C slice-by-8 tables + x^(2^i) mod P (k_decode_sections)
    hipEvent_t kb0 = nullptr, kb1 = nullptr;  // start/stop timestamps of the last k_build / k_hash_entries dispatch
    std::atomic<uint64_t> calls{0};   // construct / match parts this device has served (bsg_device_calls)
    float last_or_ms = 0.f;    // this device's last k_or_reduce_blocks (the context keeps the slowest device's: bsg_last_or_ms)
    bool or_pending = false;   // kb0/kb1 hold an un-read k_or_reduce_blocks dispatch
};

struct ArenaShard {
    uint64_t *d_words = nullptr;
    DevDesc *d_desc = nullptr;
    uint32_t n_blocks = 0;          // local blocks
    uint64_t n_words = 0;
    uint64_t max_staged_words[3] = {0, 0, 0};
    uint64_t sum_words[3] = {0, 0, 0};  // present filters, for stream-byte accounting
    uint64_t fixed_m[3] = {0, 0, 0};    // common m if all present filters share geometry, else 0
    uint32_t fixed_k[3] = {0, 0, 0};
    bool geometry_uniform[3] = {true, true, true};
};

struct Arena {
    uint32_t n_blocks = 0;
    std::vector<ArenaShard> shards;  // one per device
};

struct BatchDev {
    uint64_t *d_th = nullptr;
    uint32_t *d_prog = nullptr;
    uint32_t *d_chunk_len = nullptr;
    uint32_t *d_cw_cnt = nullptr;
    uint32_t *d_cw = nullptr;
};

struct Batch {
    uint32_t n_queries = 0;
    uint32_t Tp = 0, Wt = 0;
    uint32_t n_kinds = 0;
    uint32_t kind[3] = {0, 0, 0};
    uint32_t term_begin[3] = {0, 0, 0};
    uint32_t term_count[3] = {0, 0, 0};
    uint32_t n_chunks = 0;
    uint32_t max_depth = 1;
    uint32_t max_cw = 1;            // most verdict words any 256-query chunk references
    uint32_t Lmax = 1;              // longest lowered program of the batch (uniform chunk stride)
    bool identity_cw = false;       // few verdict words: every chunk transposes all of them, no per-chunk list
    bool many_terms = false;        // some kind has more than 128 distinct terms: k_probe_terms_many, never fused
    std::vector<BatchDev> dev;
    // A batch beyond what one launch holds (distinct terms of a kind, verdict words per 256-query chunk) is a COMPOSITE: runs of
    // its queries, each a batch of its own with only the terms its queries reference (the reference's evaluator has no such
    // limit, query_exec.go:89-126).  sub_q0[i] = first query of subs[i]; the fields above are unused then.
    std::vector<std::shared_ptr<Batch>> subs;
    std::vector<uint32_t> sub_q0;
};

struct Ingest;   // ingest_api.inc
struct ArenaStream;   // stream_api.inc

// concurrent bsg_query callers share dispatches (combine_api.inc).  No mutex on the way in: with hundreds of callers a contended
// lock hands over at ~5-10 us a time (its waiters sleep), which by itself bounds the context at ~10^5 calls a second.  The
// synchronisation — stacks of waiting calls, the collector role, the cycle slots, the wake-up tree — is host/combiner_sync.hpp.
struct Combiner {
    bsgsync::Gate sync;
    uint32_t mode = 1;            // 0: every call goes alone (bsg_set_lab key 12)
    uint32_t linger_us = 0, linger_calls = 0;   // lab (bsg_set_lab key 15): a collector waits this long / for this many queued calls
    uint32_t hot_min_queries = 8;               // an arena is streamed once for all its callers of a cycle from this many 3-term queries per 35 KB of filters per block (key 16; 0: never; scaled by the arena's bytes per block and the calls' terms: combine_api.inc).  Measured 4 / 8 / 12 / 24 on C2's arena: 4 streams too much at 64 callers over 12 arenas (5.9 vs 8.8 x 10^5), 24 leaves 64 x 10 arenas at 1.9 vs 3.0 x 10^5
    std::atomic<uint64_t> n_solo{0}, n_cycles{0}, n_cycle_calls{0}, n_dispatches{0}, n_hot{0}, max_cycle_calls{0};
    uint64_t part_bytes = 64ull << 20;           // rows of one part of a cycle (page-locked scratch, kept): a cycle beyond it is served in parts (key 24)
    uint32_t wait_spin_us = 0;                   // a collector polls its dispatch's doorbell this long, then sleeps on an event behind it (key 25); 0 = adaptive:
                                                 // 4 x the running mean of the waits that ended while polling, within [50 us, 1 ms]
    std::atomic<uint32_t> wait_ema_ns{20000};    // that mean (a cycle of 256 callers x 10 arenas runs ~150 us on an idle device, 16 callers ~15 us)
    uint32_t inline_jobs = 1;                    // a job list whose table fits the kernel arguments travels in them (key 21; 0: always uploaded)
    uint32_t profile = 0;                        // lab (key 20): callers account their own processor time (bsg_lab_query_cpu)
    std::atomic<uint64_t> n_cpu_calls{0}, ns_cpu_call{0}, ns_cpu_wait{0}, ns_cpu_duty{0}, ns_cpu_collect{0};
    std::atomic<uint64_t> ns_scatter{0}, ns_free{0}, ns_retire{0};   // parts of ns_deal
    std::atomic<uint64_t> ns_prepare{0}, ns_enqueue{0}, ns_wait{0}, ns_deal{0}, ns_wake{0};   // the combined cycles' phases, summed (collector's clock)
};

}  // namespace

struct bsg_arena_cache;   // cache_api.inc

struct bsg_ctx {
    std::shared_ptr<bsg_arena_cache> cache;   // resident file arenas across queries (cache_api.inc); the root context's
    bsg_ctx *parent = nullptr;   // non-null: this object is an error scope aliasing `parent` (bsg_scope_open)
    std::mutex err_mu;
    std::string err;             // last failure recorded on this scope / context
    std::vector<std::unique_ptr<Device>> devs;
    std::shared_mutex mu;  // handle tables (looked up under a shared lock by every probe / query call, changed under an exclusive one)
    std::map<uint64_t, std::shared_ptr<Arena>> arenas;
    std::atomic<uint64_t> arena_epoch{0};   // changes (to a value no context ever had) whenever an arena id stops naming its arena: the per-thread lookup caches of bsg_query
    std::map<uint64_t, std::shared_ptr<Batch>> batches;
    std::map<uint64_t, std::shared_ptr<Ingest>> ingests;
    std::map<uint64_t, std::shared_ptr<ArenaStream>> streams;
    uint64_t next_id = 1;
    bsg_timing timing{};
    uint32_t timed_stride = 1;   // with BSG_PROBE_TIMED, timestamp every timed_stride-th launch
    uint64_t timed_counter = 0;
    uint32_t group_limit = 1024;   // arenas one probe dispatch may cover (bsg_set_probe_group; beyond kMaxGroupArenas the records travel in device memory)
    uint32_t solo_ring_wgs = 32;  // a lone bsg_query of more workgroups than this gets its doorbell from a dispatch behind the kernel (bsg_set_lab key 22)
    uint32_t gather_cost = 256;  // a filter is gathered instead of staged when terms * k * gather_cost < its bytes
    bsg::FpKey fp_key{};         // secret key of the entries' fingerprints (drawn at bsg_open; never leaves the process)
    std::vector<uint8_t> peer;   // [i * nd + j]: 1 = device i reaches device j's memory directly (xGMI peer access enabled, or the same device)
    std::atomic<uint64_t> peer_warned{0};   // pairs (bit i * 8 + j, contexts of <= 8 devices) whose staged copies were announced
    std::vector<void *> comms;   // ncclComm_t per device (bsg_comm_init)
    uint32_t comm_world = 0, comm_rank = 0;
    uint64_t ingest_chunk_bytes = 64ull << 20;   // rows per upload chunk of bsg_ingest_rows (bsg_set_ingest_chunk)
    uint32_t spin_wait_us = 0;   // synchronous probes poll the stream this long before blocking (bsg_set_spin_wait)
    uint32_t compact_rounds = BSG_COMPACT_ROUNDS;   // many-term probe mode (bsg_set_lab)
    uint32_t load_pieces = 4;        // launches bsg_arena_load_sections splits a region's decode into, each behind its part of the copy (bsg_set_lab key 4)
    uint32_t direct_max_terms = 16;  // batches of <= 256 queries with at most this many distinct terms take k_probe_direct (bsg_set_lab key 3; 0: never)
    uint64_t bin_min_locs = 4ull << 20;             // fewer locations than this: global atomics (bsg_set_lab key 6)
    uint64_t bin_scratch_bytes = kBinScratchBytes;  // 0: bitsets beyond LDS are built with global atomics (bsg_set_lab key 2)
    uint32_t fuse_max_arenas = 4; // groups up to this many arenas ride fused (probe of group i + eval of group i-1)
    // k_probe_eval (probe + per-tile program evaluation in ONE dispatch): workgroups of a tile that share its evaluation; 0 = off,
    // the default — measured on MI355X (tools/fold_lab.py, C2, per 20 / 64 arenas): kernels 119-122 / 385 us folded vs 104 + 15.5 /
    // 322 + 39 us as two dispatches; wall 134.8 vs 133.2 us.  The survivor words written under the stream leave L2 as partial lines
    // before the other tiles complete them, which costs what the second dispatch's ramp would (bsg_set_lab key 11 turns it on)
    uint32_t fold_helpers = 0;
    uint32_t tail_split_pct = 0;   // bsg_set_lab key 19
    Combiner cmb;                // concurrent bsg_query calls merged into shared dispatches (combine_api.inc)
    // write side / matcher over several devices: a call large enough is cut into one part per device (contiguous runs of
    // filters / sets / rows), each part on a thread of its own; smaller calls take ONE device, chosen round-robin among
    // the ones whose lock is free, so independent callers (flush worker, merge, block workers) spread over the context
    std::atomic<uint32_t> next_dev{0};
    uint64_t shard_min_entries = 1ull << 18;   // bsg_hash_entries / bsg_build*: fewer entries stay on one device (bsg_set_lab key 7)
    uint64_t shard_min_row_bytes = 8ull << 20; // bsg_ingest_rows / bsg_match_rows: fewer row bytes stay on one device (bsg_set_lab key 8)
    uint32_t union_coarsen = 0;                // lab: the partitioned union starts with 2^this x too few partitions (bsg_set_lab key 10)
    uint32_t union_mode = 0;                   // file-level union: 0 = partitions in LDS (k_union_partitions), 1 = global hash tables (bsg_set_lab key 9)
    // device time of the most recent call of each family: the slowest device that took part (bsg_last_*_ms)
    float last_build_ms = 0.f, last_hash_ms = 0.f, last_decode_ms = 0.f, last_or_ms = 0.f, last_encode_ms = 0.f, last_match_ms = 0.f;
};

namespace {

void record_error(bsg_ctx *scope, const char *msg)
{
    std::lock_guard<std::mutex> lk(scope->err_mu);
    scope->err = msg;
}
bsg_ctx *root_of(bsg_ctx *c) { return c->parent ? c->parent : c; }

void free_all_ingests(bsg_ctx *ctx);   // ingest_api.inc
void destroy_comms(bsg_ctx *ctx);      // comm_api.inc
void free_all_streams(bsg_ctx *ctx);   // stream_api.inc
std::shared_ptr<bsg_arena_cache> arena_cache_create(); // cache_api.inc
void arena_cache_destroy(bsg_ctx *ctx);
int32_t ensure_lower_table(Device &d);   // ingest_api.inc: the unicode.ToLower table the walkers fold with

struct SectionsOut { uint8_t *region; uint64_t cap; uint64_t *sec_off; };   // encode_api.inc
uint64_t section_len(const bsg_filter_desc *d3);
int32_t encode_sections_device(Device &d, const uint64_t *d_words, const bsg_filter_desc *desc, uint32_t n_blocks,
                               uint8_t *out_region, uint64_t region_cap, uint64_t *out_sec_off, float *ms);

int32_t use_device(Device &d)
{
    HIP_TRY(hipSetDevice(d.id));
    return BSG_OK;
}

// Device-to-device copy between two entries of the context, on `stream` (a stream of either device).  The same physical device:
// a plain copy.  A pair with peer access: hipMemcpyPeerAsync over xGMI.  A pair WITHOUT it: the same call — the runtime stages
// the bytes through host memory — announced on stderr once per pair (BSG_QUIET=1 silences it), because a multi-GPU flush or
// merge that runs at PCIe speed should not look like a slow kernel.  bsg_peer_access reports the matrix.
uint32_t dev_index(const bsg_ctx *ctx, const Device *d)
{
    for (uint32_t i = 0; i < ctx->devs.size(); ++i) if (ctx->devs[i].get() == d) return i;
    return 0;
}

hipError_t peer_copy(bsg_ctx *ctx, uint32_t dst_i, void *dst, uint32_t src_i, const void *src, size_t bytes, hipStream_t stream)
{
    Device &D = *ctx->devs[dst_i], &S = *ctx->devs[src_i];
    if (D.id == S.id) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    const uint32_t nd = (uint32_t)ctx->devs.size();
    const bool direct = ctx->peer.size() == (size_t)nd * nd && ctx->peer[(size_t)dst_i * nd + src_i] && ctx->peer[(size_t)src_i * nd + dst_i];
    if (!direct && nd <= 8) {
        const uint64_t bit = 1ull << (dst_i * 8 + src_i);
        if (!(ctx->peer_warned.fetch_or(bit) & bit) && !getenv("BSG_QUIET"))
            fprintf(stderr, "[bloomgpu] no peer access between devices %d and %d: copies between them are staged through host memory\n", S.id, D.id);
    }
    return hipMemcpyPeerAsync(dst, D.id, src, S.id, bytes, stream);
}

// The device for a call that stays on ONE device of the context: round-robin, preferring a device nobody holds right now
// (the flush worker, a merge and the block workers of concurrent queries then land on different GPUs).  The answer is a
// hint — the caller takes the device's lock the usual way.
uint32_t pick_device(bsg_ctx *ctx)
{
    const uint32_t nd = (uint32_t)ctx->devs.size();
    if (nd == 1) return 0;
    const uint32_t start = ctx->next_dev.fetch_add(1, std::memory_order_relaxed) % nd;
    for (uint32_t i = 0; i < nd; ++i) {
        const uint32_t di = (start + i) % nd;
        if (ctx->devs[di]->mu.try_lock()) { ctx->devs[di]->mu.unlock(); return di; }
    }
    return start;
}

// part(i) for i < n, parts 1 .. n-1 on threads of their own and part 0 on the caller's; every thread reports into the
// error scope the call was made on.  First failing part's status.
template <class F>
int32_t run_parts(uint32_t n, F &&part)
{
    if (n == 0) return BSG_OK;
    if (n == 1) return part(0u);
    std::vector<int32_t> rc(n, BSG_OK);
    std::vector<std::string> msg(n);          // a worker's message lives in ITS thread-local g_err: carried back by value
    bsg_ctx *scope = tl_scope;
    std::vector<std::thread> th;
    th.reserve(n - 1);
    for (uint32_t i = 1; i < n; ++i)
        th.emplace_back([&rc, &msg, &part, scope, i]() {
            tl_scope = scope;
            g_err.clear();
            rc[i] = part(i);
            if (rc[i]) msg[i] = g_err;
            tl_scope = nullptr;
        });
    g_err.clear();
    rc[0] = part(0u);
    if (rc[0]) msg[0] = g_err;
    for (auto &t : th) t.join();
    // the first failing part's code AND message, re-issued on the caller's thread: bsg_last_error(NULL), the wrappers that save
    // g_err around a nested call, and the scope's slot (which a later part may have overwritten) all name the same failure
    for (uint32_t i = 0; i < n; ++i) if (rc[i]) return fail(rc[i], "%s", msg[i].empty() ? "a device part failed" : msg[i].c_str());
    return BSG_OK;
}

// cuts [0, n) items with the given costs into at most `parts` contiguous runs of about equal cost; returns the run
// boundaries (size runs + 1).  `unit`: boundaries fall on multiples of it (3 = whole blocks of filters).
std::vector<uint32_t> balanced_cuts(const std::vector<uint64_t> &cost, uint32_t parts, uint32_t unit = 1)
{
    const uint32_t n = (uint32_t)cost.size();
    std::vector<uint32_t> cuts{0};
    uint64_t total = 0;
    for (uint64_t c : cost) total += c;
    uint64_t acc = 0;
    uint32_t made = 1;
    for (uint32_t i = 0; i < n && made < parts; ++i) {
        acc += cost[i];
        if ((i + 1) % unit == 0 && i + 1 < n && acc * parts >= total * made) { cuts.push_back(i + 1); ++made; }
    }
    cuts.push_back(n);
    return cuts;
}

void free_arena(bsg_ctx *ctx, Arena &a)
{
    for (size_t i = 0; i < a.shards.size(); ++i) {
        (void)hipSetDevice(ctx->devs[i]->id);
        if (a.shards[i].d_words) (void)hipFree(a.shards[i].d_words);
        if (a.shards[i].d_desc) (void)hipFree(a.shards[i].d_desc);
    }
}

void free_batch(bsg_ctx *ctx, Batch &b)
{
    for (auto &sub : b.subs) free_batch(ctx, *sub);
    b.subs.clear();
    for (size_t i = 0; i < b.dev.size(); ++i) {
        (void)hipSetDevice(ctx->devs[i]->id);
        if (b.dev[i].d_th) (void)hipFree(b.dev[i].d_th);       // one block holds the five tables (probe_api.inc: BatchTables), d_th is its base
        b.dev[i] = BatchDev{};
    }
}

int32_t drain_timing(bsg_ctx *ctx, Device &d)
{
    if (d.pending.empty()) return BSG_OK;
    HIP_TRY(hipStreamSynchronize(d.stream));
    for (auto &t : d.pending) {
        float a = 0, b = 0;
        if (t.has_k1) HIP_TRY(hipEventElapsedTime(&a, t.k1s, t.k1e));
        if (t.has_k2) HIP_TRY(hipEventElapsedTime(&b, t.k2s, t.k2e));
        std::lock_guard<std::shared_mutex> lk(ctx->mu);
        if (t.has_k1 && t.folded) {
            ctx->timing.n_folded += 1;
            ctx->timing.ms_folded_kernel += a;
            ctx->timing.folded_stream_bytes += t.bytes;
            ctx->timing.n_folded_arenas += t.n_arenas;
        } else if (t.has_k1 && t.fused) {
            ctx->timing.n_fused += 1;
            ctx->timing.ms_fused_kernel += a;
            ctx->timing.fused_stream_bytes += t.bytes;
            ctx->timing.n_fused_arenas += t.n_arenas;
        } else if (t.has_k1) {
            ctx->timing.n_probes += 1;
            ctx->timing.ms_terms_kernel += a;
            ctx->timing.stream_bytes += t.bytes;
            ctx->timing.n_probe_arenas += t.n_arenas;
        }
        if (t.has_k2) { ctx->timing.n_eval += 1; ctx->timing.ms_eval_kernel += b; }
        d.free_events.push_back(t);
    }
    d.pending.clear();
    return BSG_OK;
}

// ---- program lowering: public n-ary postfix -> internal binary postfix ----
struct Node {
    uint32_t opc;   // BSG_OP_*
    uint32_t arg;   // TERM: verdict position
    std::vector<uint32_t> kids;
};

void emit_node(const std::vector<Node> &nodes, uint32_t id, std::vector<uint32_t> &out)
{
    const Node &n = nodes[id];
    switch (n.opc) {
    case BSG_OP_TERM: out.push_back((0u << 28) | n.arg); return;
    case BSG_OP_TRUE: out.push_back(3u << 28); return;
    case BSG_OP_FALSE: out.push_back(4u << 28); return;
    default: break;
    }
    if (n.kids.empty()) {  // And() == true, Or() == false (query_exec.go:105-121)
        out.push_back((n.opc == BSG_OP_AND ? 3u : 4u) << 28);
        return;
    }
    emit_node(nodes, n.kids[0], out);
    for (size_t c = 1; c < n.kids.size(); ++c) {
        emit_node(nodes, n.kids[c], out);
        out.push_back((n.opc == BSG_OP_AND ? 1u : 2u) << 28);
    }
}

// Returns BSG_OK and the lowered program, or BSG_E_INVALID.
int32_t lower_program(const uint32_t *ops, uint32_t n_ops, uint32_t n_terms, const std::vector<uint32_t> &term_pos,
                      std::vector<uint32_t> &out, uint32_t &depth)
{
    out.clear();
    depth = 1;
    if (n_ops == 0) return BSG_OK;  // nil query: true
    std::vector<Node> nodes;
    std::vector<uint32_t> stack;
    nodes.reserve(n_ops);
    for (uint32_t j = 0; j < n_ops; ++j) {
        const uint32_t opc = ops[j] >> 28, arg = ops[j] & 0x0FFFFFFFu;
        Node n{opc, 0, {}};
        switch (opc) {
        case BSG_OP_TERM:
            if (arg >= n_terms) return fail(BSG_E_INVALID, "program references term %u of %u", arg, n_terms);
            n.arg = term_pos[arg];
            break;
        case BSG_OP_AND:
        case BSG_OP_OR:
            if (arg > stack.size()) return fail(BSG_E_INVALID, "program op %u pops %u of %zu", j, arg, stack.size());
            n.kids.assign(stack.end() - arg, stack.end());
            stack.resize(stack.size() - arg);
            break;
        case BSG_OP_TRUE:
        case BSG_OP_FALSE:
            break;
        default:
            return fail(BSG_E_INVALID, "unknown program opcode %u", opc);
        }
        nodes.push_back(std::move(n));
        stack.push_back((uint32_t)nodes.size() - 1);
    }
    if (stack.size() != 1) return fail(BSG_E_INVALID, "program leaves %zu values on the stack", stack.size());
    emit_node(nodes, stack[0], out);
    uint32_t sp = 0;
    for (uint32_t op : out) {
        const uint32_t opc = op >> 28;
        if (opc == 1u || opc == 2u) --sp; else ++sp;
        depth = std::max(depth, sp);
    }
    return BSG_OK;
}

int32_t get_arena(bsg_ctx *ctx, uint64_t id, std::shared_ptr<Arena> &out)
{
    std::shared_lock<std::shared_mutex> lk(ctx->mu);
    auto it = ctx->arenas.find(id);
    if (it == ctx->arenas.end()) return fail(BSG_E_NOTFOUND, "unknown arena id %llu", (unsigned long long)id);
    out = it->second;
    return BSG_OK;
}

// bsg_query's arena lookup.  With hundreds of caller threads the shared lock above and the arena's reference count are two cache
// lines every call writes — ~1 us each when 256 threads on two sockets take turns (measured: a call's whole preparation 0.23 us on one
// thread, 12 us on 256).  So a thread remembers the arenas it looked up: ids are never reused (an id names one arena or nothing), so
// an entry stays valid until SOME id of the context is freed — arena_epoch then takes a value no context ever had, and the thread's
// next lookup forgets everything.  What the caller gets is an alias of the cached pointer that counts on a thread-private block.
std::atomic<uint64_t> g_arena_epoch{1};
struct ArenaCache {
    const bsg_ctx *ctx = nullptr;
    uint64_t epoch = 0;
    std::map<uint64_t, std::shared_ptr<Arena>> arenas;
    std::shared_ptr<int> owner = std::make_shared<int>(0);
};
// The returned pointer is a NON-OWNING alias of the entry in this thread's cache: valid for the duration of the calling bsg_query
// only (QReq::arenas must not outlive the call).  The cache holds host objects, never device memory (bsg_arena_free frees that at
// once); a thread keeps at most 4 096 of them and drops them all when any arena id of its context stops naming its arena (epoch).
int32_t get_arena_cached(bsg_ctx *ctx, uint64_t id, std::shared_ptr<Arena> &out)
{
    thread_local ArenaCache cache;
    const uint64_t epoch = ctx->arena_epoch.load(std::memory_order_acquire);
    if (cache.ctx != ctx || cache.epoch != epoch) { cache.arenas.clear(); cache.ctx = ctx; cache.epoch = epoch; }
    auto it = cache.arenas.find(id);
    if (it == cache.arenas.end()) {
        std::shared_ptr<Arena> a;
        if (int32_t rc = get_arena(ctx, id, a)) return rc;
        if (cache.arenas.size() >= 4096) cache.arenas.clear();
        it = cache.arenas.emplace(id, std::move(a)).first;
        // (an id freed between the epoch read and this lookup fails in get_arena; one freed after it is the caller's race, as before)
    }
    out = std::shared_ptr<Arena>(cache.owner, it->second.get());
    return BSG_OK;
}

int32_t get_batch(bsg_ctx *ctx, uint64_t id, std::shared_ptr<Batch> &out)
{
    std::shared_lock<std::shared_mutex> lk(ctx->mu);
    auto it = ctx->batches.find(id);
    if (it == ctx->batches.end()) return fail(BSG_E_NOTFOUND, "unknown batch id %llu", (unsigned long long)id);
    out = it->second;
    return BSG_OK;
}

int32_t validate_descs(const bsg_filter_desc *desc, size_t n, uint64_t n_words)
{
    for (size_t i = 0; i < n; ++i) {
        if (desc[i].m == 0) continue;
        const uint64_t nw = (desc[i].m + 63) / 64;
        if (desc[i].k == 0 || desc[i].k > kMaxHashCount)
            return fail(BSG_E_INVALID, "descriptor %zu: k = %u outside [1, %llu]", i, desc[i].k, (unsigned long long)kMaxHashCount);
        if (desc[i].word_off > n_words || nw > n_words - desc[i].word_off)
            return fail(BSG_E_INVALID, "descriptor %zu: words [%llu, +%llu) outside arena of %llu words", i,
                        (unsigned long long)desc[i].word_off, (unsigned long long)nw, (unsigned long long)n_words);
    }
    return BSG_OK;
}

}  // namespace

extern "C" {

int32_t bsg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t bsg_open(const int32_t *device_ids, int32_t n_devices, bsg_ctx **out_ctx)
{
    if (!out_ctx) return fail(BSG_E_INVALID, "out_ctx is null");
    *out_ctx = nullptr;
    if (n_devices < 1 || !device_ids) return fail(BSG_E_INVALID, "need at least one device id");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(BSG_E_NODEVICE, "no HIP device visible (libbloomgpu has no CPU fallback)");
    auto ctx = std::make_unique<bsg_ctx>();
    ctx->cache = arena_cache_create();
    ctx->arena_epoch.store(g_arena_epoch.fetch_add(1, std::memory_order_relaxed), std::memory_order_relaxed);
    {
        std::random_device rd;
        auto r64 = [&]() { return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)(uintptr_t)ctx.get() << 7) ^
                                  (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(); };
        ctx->fp_key = bsg::FpKey{r64(), r64() | 1, r64() | 1};
    }
    for (int32_t i = 0; i < n_devices; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= count)
            return fail(BSG_E_INVALID, "device id %d out of range [0,%d)", device_ids[i], count);
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_ids[i]));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(BSG_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_ids[i],
                        prop.gcnArchName);
        auto d = std::make_unique<Device>();
        d->id = device_ids[i];
        d->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
        HIP_TRY(hipSetDevice(d->id));
        HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        // more than 64 KiB of dynamic LDS per workgroup is opt-in per kernel
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_probe_terms), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_probe_terms_many), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_probe_terms_ext), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_probe_terms_many_ext), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_probe_fused), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_probe_eval), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_build), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_build_sets), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget - bsg::kSetListBytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(bsg::k_union_partitions), hipFuncAttributeMaxDynamicSharedMemorySize, bsg::kPartLdsBytes));

        ctx->devs.push_back(std::move(d));
    }
    // shards exchange partial bitsets, partial file-level sets and resident arenas device-to-device: enable xGMI peer access
    // where the pair allows it and REMEMBER where it does not — a copy between such a pair still works (the runtime stages it
    // through host memory) but at PCIe speed, and peer_copy says so once per pair instead of being silently slow
    const uint32_t nd_open = (uint32_t)ctx->devs.size();
    ctx->peer.assign((size_t)nd_open * nd_open, 1);
    for (uint32_t i = 0; i < nd_open; ++i)
        for (uint32_t j = 0; j < nd_open; ++j) {
            Device &a = *ctx->devs[i], &b = *ctx->devs[j];
            if (a.id == b.id) continue;
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, a.id, b.id);
            if (e == hipSuccess && can) {
                (void)hipSetDevice(a.id);
                e = hipDeviceEnablePeerAccess(b.id, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            }
            if (e != hipSuccess || !can) {
                (void)hipGetLastError();
                ctx->peer[(size_t)i * nd_open + j] = 0;
            }
        }
    *out_ctx = ctx.release();
    return BSG_OK;
}

int32_t bsg_open_err(const int32_t *device_ids, int32_t n_devices, bsg_ctx **out_ctx, char *errbuf, uint64_t cap)
{
    const int32_t rc = bsg_open(device_ids, n_devices, out_ctx);
    if (errbuf && cap) {
        const std::string &m = rc ? g_err : std::string();
        const size_t n = std::min<size_t>(m.size(), cap - 1);
        memcpy(errbuf, m.data(), n);
        errbuf[n] = 0;
    }
    return rc;
}

int32_t bsg_close(bsg_ctx *ctx)
{
    if (!ctx) return BSG_OK;
    if (ctx->parent) { delete ctx; return BSG_OK; }   // an error scope owns nothing but its message
    for (auto &kv : ctx->arenas) free_arena(ctx, *kv.second);
    for (auto &kv : ctx->batches) free_batch(ctx, *kv.second);
    free_all_ingests(ctx);
    free_all_streams(ctx);
    destroy_comms(ctx);
    for (auto &dp : ctx->devs) {
        Device &d = *dp;
        (void)hipSetDevice(d.id);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.copy_stream) {
            (void)hipStreamSynchronize(d.copy_stream);
            for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(d.ev_eval[i]); (void)hipEventDestroy(d.ev_copy[i]); }
            (void)hipStreamDestroy(d.copy_stream);
        }
        if (d.kb0) { (void)hipEventDestroy(d.kb0); (void)hipEventDestroy(d.kb1); }
        if (d.aux_stream) { (void)hipEventDestroy(d.ev_aux[0]); (void)hipEventDestroy(d.ev_aux[1]); (void)hipStreamDestroy(d.aux_stream); }
        d.pool.trim(0);
        for (uint8_t *p : d.idle_staging) (void)hipHostFree(p);
        for (auto &k : d.idle_kits) {
            (void)hipStreamDestroy(k.copy); (void)hipEventDestroy(k.copied);
            (void)hipEventDestroy(k.staged_free[0]); (void)hipEventDestroy(k.staged_free[1]);
        }
        for (auto &p : d.direct_bufs) (void)hipHostFree(p.first);
        for (auto &te : d.table_cache) te.buf.release();
        for (auto &cs : d.cmb_sets) {
            if (cs.d_block) (void)hipFree(cs.d_block);
            if (cs.stage.first) (void)hipHostFree(cs.stage.first);
            if (cs.result.first) (void)hipHostFree(cs.result.first);
            if (cs.drained) (void)hipEventDestroy(cs.drained);
        }
        if (d.d_direct_count) (void)hipFree(d.d_direct_count);
        if (d.d_crc) (void)hipFree(d.d_crc);
        if (d.d_lower) (void)hipFree(d.d_lower);
        for (auto *v : {&d.pending, &d.free_events})
            for (auto &t : *v) {
                (void)hipEventDestroy(t.k1s); (void)hipEventDestroy(t.k1e);
                (void)hipEventDestroy(t.k2s); (void)hipEventDestroy(t.k2e);
            }
        d.V[0].release(); d.V[1].release(); d.out[0].release(); d.out[1].release(); d.stage_a.release(); d.stage_off.release(); d.stage_h.release();
        d.stage_fstart.release(); d.stage_desc.release(); d.stage_items.release(); d.stage_words.release(); d.stage_region.release();
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    arena_cache_destroy(ctx);
    delete ctx;
    return BSG_OK;
}

const char *bsg_last_error(bsg_ctx *ctx)
{
    if (!ctx) return g_err.c_str();
    {
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        g_ret = ctx->err;
    }
    return g_ret.c_str();
}

int32_t bsg_last_error_copy(bsg_ctx *ctx, char *buf, uint64_t cap)
{
    if (!buf || cap == 0) return BSG_E_INVALID;
    std::string m;
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); m = ctx->err; }
    else m = g_err;
    const size_t n = std::min<size_t>(m.size(), cap - 1);
    memcpy(buf, m.data(), n);
    buf[n] = 0;
    return BSG_OK;
}

int32_t bsg_scope_open(bsg_ctx *ctx, bsg_ctx **out_scope)
{
    if (!ctx || !out_scope) return fail(BSG_E_INVALID, "null argument");
    auto *s = new bsg_ctx();
    s->parent = root_of(ctx);
    *out_scope = s;
    return BSG_OK;
}

int32_t bsg_sync(bsg_ctx *ctx)
{
    BSG_ENTER(ctx);
    if (!ctx) return fail(BSG_E_INVALID, "ctx is null");
    for (auto &dp : ctx->devs) {
        std::lock_guard<std::mutex> lk(dp->mu);
        if (int32_t rc = use_device(*dp)) return rc;
        // (measured and dropped: hipStreamWriteValue64 into page-locked memory + polling instead — the stream memory op itself
        // costs more than the wait it saves: 8.9 vs 8.2 us per step over a 20-arena call; only a kernel's own doorbell pays,
        // see k_probe_direct)
        HIP_TRY(hipStreamSynchronize(dp->stream));
        if (dp->copy_stream) HIP_TRY(hipStreamSynchronize(dp->copy_stream));
        dp->copy_busy[0] = dp->copy_busy[1] = false;
    }
    return BSG_OK;
}

int32_t bsg_estimate_parameters(uint64_t n, double p, uint64_t *m, uint64_t *k)
{
    if (!m || !k) return fail(BSG_E_INVALID, "null output");
    if (n == 0 || !(p > 0.0 && p < 1.0)) return fail(BSG_E_INVALID, "need n >= 1 and 0 < p < 1");
    // bloom/v3 EstimateParameters; New() clamps both to >= 1.
    const double mm = std::ceil(-1.0 * (double)n * std::log(p) / std::pow(std::log(2.0), 2.0));
    const double kk = std::ceil(std::log(2.0) * mm / (double)n);
    *m = mm < 1.0 ? 1 : (uint64_t)mm;
    *k = kk < 1.0 ? 1 : (uint64_t)kk;
    return BSG_OK;
}

// entries [e0, e1) on one device
static int32_t hash_entries_on(Device &d, const uint8_t *bytes, const uint32_t *offsets, uint32_t e0, uint32_t e1, uint64_t *out_h, float *ms)
{
    const uint32_t n = e1 - e0, b0 = offsets[e0], n_bytes = offsets[e1] - b0;
    d.calls.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(d.mu);
    if (int32_t rc = use_device(d)) return rc;
    HIP_TRY(d.stage_a.reserve((size_t)n_bytes + 64));
    HIP_TRY(d.stage_off.reserve((size_t)n + 1));
    HIP_TRY(d.stage_h.reserve((size_t)n * 4));
    if (n_bytes) HIP_TRY(hipMemcpyAsync(d.stage_a.p, bytes + b0, n_bytes, hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.stage_off.p, offsets + e0, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, d.stream));
    if (!d.kb0) { HIP_TRY(hipEventCreate(&d.kb0)); HIP_TRY(hipEventCreate(&d.kb1)); }
    // the offsets stay absolute: the byte pointer is moved back by the run's first offset instead (never dereferenced below the buffer)
    hipExtLaunchKernelGGL(bsg::k_hash_entries, dim3((n + 255) / 256), dim3(256), 0, d.stream, d.kb0, d.kb1, 0,
                          (const uint8_t *)d.stage_a.p - b0, (const uint32_t *)d.stage_off.p, n, d.stage_h.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_h + (size_t)e0 * 4, d.stage_h.p, (size_t)n * 32, hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    HIP_TRY(hipEventElapsedTime(ms, d.kb0, d.kb1));
    return BSG_OK;
}

int32_t bsg_hash_entries(bsg_ctx *ctx, const uint8_t *bytes, const uint32_t *offsets, uint32_t n_entries,
                         uint64_t *out_h)
{
    BSG_ENTER(ctx);
    if (!ctx) return fail(BSG_E_INVALID, "ctx is null");
    if (n_entries == 0) return BSG_OK;
    if (!offsets || !out_h) return fail(BSG_E_INVALID, "null argument");
    for (uint32_t e = 0; e < n_entries; ++e)
        if (offsets[e + 1] < offsets[e]) return fail(BSG_E_INVALID, "offsets not monotone at %u", e);
    const uint32_t n_bytes = offsets[n_entries];
    if (n_bytes && !bytes) return fail(BSG_E_INVALID, "bytes is null");
    // a large batch is cut into one contiguous run of entries per device; a small one takes one device
    const uint32_t nd = (uint32_t)ctx->devs.size();
    const uint32_t parts = (nd > 1 && n_entries >= ctx->shard_min_entries) ? std::min<uint32_t>(nd, n_entries) : 1;
    std::vector<float> ms(parts, 0.f);
    const uint32_t first = parts == 1 ? pick_device(ctx) : 0;
    const int32_t rc = run_parts(parts, [&](uint32_t i) -> int32_t {
        const uint32_t e0 = (uint32_t)((uint64_t)n_entries * i / parts), e1 = (uint32_t)((uint64_t)n_entries * (i + 1) / parts);
        return hash_entries_on(*ctx->devs[(first + i) % nd], bytes, offsets, e0, e1, out_h, &ms[i]);
    });
    if (rc) return rc;
    std::lock_guard<std::shared_mutex> lk(ctx->mu);
    ctx->last_hash_ms = *std::max_element(ms.begin(), ms.end());
    return BSG_OK;
}

// One bitset beyond LDS from binned locations (bin_build.hip.h): a.t / a.d / a.n_slots / a.n_locs_cap / a.out / a.overflow
// come filled in; the scratch this takes from the pool is appended to `scratch` (freed by the caller after the stream has
// drained).  first / last: timestamps of the first and the last dispatch, nullptr for none.
static int32_t enqueue_binned_build(Device &d, bsg::BinArgs a, bool dense, std::vector<void *> &scratch, hipEvent_t first, hipEvent_t last)
{
    a.n_windows = (uint32_t)((a.d.m + bsg::kBinWindowBits - 1) / bsg::kBinWindowBits);
    HIP_TRY(d.pool.alloc((void **)&a.prefix, ((size_t)a.n_windows + 1) * 4));
    scratch.push_back(a.prefix);
    HIP_TRY(d.pool.alloc((void **)&a.cursor, (size_t)a.n_windows * 4));
    scratch.push_back(a.cursor);
    HIP_TRY(d.pool.alloc((void **)&a.locs, std::max<size_t>(a.n_locs_cap, 1) * 4));
    scratch.push_back(a.locs);
    HIP_TRY(hipMemsetAsync(a.prefix, 0, ((size_t)a.n_windows + 1) * 4, d.stream));
    const uint32_t tiles = (uint32_t)((a.n_slots + bsg::kBinTile - 1) / bsg::kBinTile);
    if (dense) hipExtLaunchKernelGGL((bsg::k_bin_pass<false, true>), dim3(tiles), dim3(bsg::kBinThreads), 0, d.stream, first, nullptr, 0, a);
    else       hipExtLaunchKernelGGL((bsg::k_bin_pass<false, false>), dim3(tiles), dim3(bsg::kBinThreads), 0, d.stream, first, nullptr, 0, a);
    hipLaunchKernelGGL(bsg::k_bin_scan, dim3(1), dim3(bsg::kBinThreads), 0, d.stream, a);
    if (dense) hipLaunchKernelGGL((bsg::k_bin_pass<true, true>), dim3(tiles), dim3(bsg::kBinThreads), 0, d.stream, a);
    else       hipLaunchKernelGGL((bsg::k_bin_pass<true, false>), dim3(tiles), dim3(bsg::kBinThreads), 0, d.stream, a);
    hipExtLaunchKernelGGL(bsg::k_bin_apply, dim3(a.n_windows), dim3(bsg::kBinThreads), 0, d.stream, nullptr, last, 0, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BSG_E_HIP, "binned build: %s", hipGetErrorString(e));
    return BSG_OK;
}
// Binning pays from a few million locations on (three more launches, scratch from the pool); a bitset just beyond LDS with
// a few ten thousand entries stays L2-resident under its atomics (1 000 such filters in one call: one k_build launch
// instead of 5 000 small ones).
static bool binned_build_fits(const bsg_ctx *root, uint64_t m, uint64_t n_entries, uint64_t k)
{
    const uint64_t n_locs = n_entries * k;
    return m < (1ull << 31) && n_locs >= std::max<uint64_t>(root->bin_min_locs, 1) && n_locs < (1ull << 32) - 4096 &&
           n_locs * 4 <= root->bin_scratch_bytes;
}

// One part of a build: filters [f0, f1) — whose entries [fstart[f0], fstart[f1]) are contiguous — on one device.  The
// part's filters occupy words [w_lo, w_hi) of the caller's arena (disjoint from every other part's); sections: the part's
// blocks f0 / 3 .. f1 / 3 are serialised at region + region_off (their offsets, relative to the part, into sec_off_local).
struct BuildPart {
    uint32_t f0 = 0, f1 = 0;
    uint64_t w_lo = 0, w_hi = 0;
    uint64_t region_off = 0, region_len = 0;
    std::vector<uint64_t> sec_off_local;
    float ms = 0.f, encode_ms = 0.f;
};

static int32_t build_on_device(bsg_ctx *ctx, Device &d, BuildPart &P, const uint8_t *bytes, const uint32_t *offsets, const uint64_t *h,
                               const uint32_t *fstart, const bsg_filter_desc *desc, uint64_t *out_words, const SectionsOut *sections)
{
    const uint32_t f0 = P.f0, f1 = P.f1, nf = f1 - f0;
    const uint32_t e0 = fstart[f0], e1 = fstart[f1], ne_all = e1 - e0;
    const uint32_t b0 = (!h && ne_all) ? offsets[e0] : 0, n_bytes = (!h && ne_all) ? offsets[e1] - b0 : 0;
    const uint64_t n_words = std::max<uint64_t>(P.w_hi - P.w_lo, 2);
    std::vector<DevDesc> dd(nf);
    std::vector<bsg_filter_desc> local(nf);          // the part's descriptors with word offsets relative to the part
    std::vector<bsg::BuildItem> items;
    std::vector<uint32_t> binned;                  // bitsets beyond LDS: assembled window by window (bin_build.hip.h)
    uint64_t max_staged = 0;
    for (uint32_t f = f0; f < f1; ++f) {
        local[f - f0] = desc[f];
        if (desc[f].m) local[f - f0].word_off = desc[f].word_off - P.w_lo;
        dd[f - f0] = DevDesc{local[f - f0].word_off, desc[f].m, barrett_magic(desc[f].m), desc[f].k, 0};
        if (desc[f].m == 0) continue;
        const uint64_t nw = (desc[f].m + 63) / 64;
        if (nw <= kLdsCapWords) {
            items.push_back({f, fstart[f], fstart[f + 1], 1u});
            max_staged = std::max(max_staged, nw);
        } else if (binned_build_fits(ctx, desc[f].m, fstart[f + 1] - fstart[f], desc[f].k)) {
            binned.push_back(f);
        } else {
            for (uint32_t e = fstart[f]; e < fstart[f + 1]; e += kBuildSliceEntries)
                items.push_back({f, e, std::min(fstart[f + 1], e + kBuildSliceEntries), 0u});
        }
    }
    // Longest first: workgroups are handed out in item order, so the few-entry items (a block's field filter: nine entries) fill the
    // tail of the launch instead of taking slots between the long ones.
    if (items.size() > 1)
        std::stable_sort(items.begin(), items.end(), [](const bsg::BuildItem &x, const bsg::BuildItem &y) { return x.e_end - x.e_begin > y.e_end - y.e_begin; });
    d.calls.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(d.mu);
    if (int32_t rc = use_device(d)) return rc;
    HIP_TRY(d.stage_words.reserve(n_words));
    HIP_TRY(hipMemsetAsync(d.stage_words.p, 0, n_words * 8, d.stream));
    bool launched = false;
    std::vector<void *> scratch;
    struct ScratchGuard { Device &d; std::vector<void *> &v; ~ScratchGuard() { if (!v.empty()) (void)hipStreamSynchronize(d.stream); for (void *p : v) d.pool.free(p); } } sguard{d, scratch};
    uint32_t *d_over = nullptr;
    if (!items.empty() || !binned.empty()) {
        HIP_TRY(d.stage_desc.reserve(nf));
        HIP_TRY(d.stage_items.reserve(std::max<size_t>(items.size(), 1)));
        HIP_TRY(hipMemcpyAsync(d.stage_desc.p, dd.data(), dd.size() * sizeof(DevDesc), hipMemcpyHostToDevice, d.stream));
        if (!items.empty())
            HIP_TRY(hipMemcpyAsync(d.stage_items.p, items.data(), items.size() * sizeof(bsg::BuildItem), hipMemcpyHostToDevice, d.stream));
        // Items keep the caller's ABSOLUTE entry and filter indices; the device arrays hold the part's run only, so their
        // base pointers are moved back by the run's first index (never dereferenced below the buffers).
        bsg::BuildArgs a{};
        if (h) {
            HIP_TRY(d.stage_h.reserve((size_t)std::max(ne_all, 1u) * 4));
            if (ne_all) HIP_TRY(hipMemcpyAsync(d.stage_h.p, h + (size_t)e0 * 4, (size_t)ne_all * 32, hipMemcpyHostToDevice, d.stream));
            a.h = d.stage_h.p - (size_t)e0 * 4;
        } else {
            HIP_TRY(d.stage_a.reserve((size_t)n_bytes + 64));
            HIP_TRY(d.stage_off.reserve((size_t)ne_all + 1));
            if (n_bytes) HIP_TRY(hipMemcpyAsync(d.stage_a.p, bytes + b0, n_bytes, hipMemcpyHostToDevice, d.stream));
            if (ne_all)
                HIP_TRY(hipMemcpyAsync(d.stage_off.p, offsets + e0, ((size_t)ne_all + 1) * 4, hipMemcpyHostToDevice, d.stream));
            a.bytes = d.stage_a.p - b0;
            a.off = d.stage_off.p - e0;
            if (!binned.empty()) HIP_TRY(d.stage_h.reserve((size_t)std::max(ne_all, 1u) * 4));   // the binned filters' entries are hashed once, up front
        }
        a.items = d.stage_items.p;
        a.desc = d.stage_desc.p - f0;
        a.out = d.stage_words.p;
        const size_t lds = std::max<uint64_t>(max_staged, 2) * 8;
        if (!d.kb0) { HIP_TRY(hipEventCreate(&d.kb0)); HIP_TRY(hipEventCreate(&d.kb1)); }
        // d.kb0 = start of the first dispatch, d.kb1 = end of the last one
        auto first_ev = [&]() { hipEvent_t s = launched ? nullptr : d.kb0; launched = true; return s; };
        if (!items.empty()) {
            hipExtLaunchKernelGGL(bsg::k_build, dim3((uint32_t)items.size()), dim3(bsg::kBuildThreads), (uint32_t)lds, d.stream,
                                  first_ev(), binned.empty() ? d.kb1 : nullptr, 0, a);
            HIP_TRY(hipGetLastError());
        }
        if (!binned.empty()) {
            HIP_TRY(d.pool.alloc((void **)&d_over, 64));
            scratch.push_back(d_over);
            HIP_TRY(hipMemsetAsync(d_over, 0, 64, d.stream));
        }
        for (size_t bi = 0; bi < binned.size(); ++bi) {
            const uint32_t f = binned[bi];
            const uint32_t fe0 = fstart[f], ne = fstart[f + 1] - fstart[f];
            uint64_t *hslot = d.stage_h.p + (size_t)(fe0 - e0) * 4;     // this filter's hashes inside the part's staging
            if (!h) {
                hipExtLaunchKernelGGL(bsg::k_hash_entries, dim3((ne + 255) / 256), dim3(256), 0, d.stream, first_ev(), nullptr, 0,
                                      (const uint8_t *)d.stage_a.p - b0, (const uint32_t *)d.stage_off.p + (fe0 - e0), ne, hslot);
                HIP_TRY(hipGetLastError());
            }
            bsg::BinArgs b{};
            b.t = bsg::IngestTable{hslot, nullptr, 0, 0};      // (a dense list of hashes: never inserted into)
            b.d = dd[f - f0];
            b.n_slots = ne;
            b.n_locs_cap = (uint32_t)((uint64_t)ne * desc[f].k);
            b.overflow = d_over;
            b.out = d.stage_words.p;
            if (int32_t rc = enqueue_binned_build(d, b, true, scratch, first_ev(), bi + 1 == binned.size() ? d.kb1 : nullptr)) return rc;
        }
    }
    if (sections) {      // the words never leave the device: encodeFilterSection runs there too
        P.sec_off_local.assign((size_t)nf / 3 + 1, 0);
        if (int32_t rc = encode_sections_device(d, d.stage_words.p, local.data(), nf / 3, sections->region + P.region_off, P.region_len,
                                                P.sec_off_local.data(), &P.encode_ms)) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(out_words + P.w_lo, d.stage_words.p, (P.w_hi - P.w_lo) * 8, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
    }
    P.ms = 0.f;
    if (launched) HIP_TRY(hipEventElapsedTime(&P.ms, d.kb0, d.kb1));
    return BSG_OK;
}

static int32_t build_common(bsg_ctx *ctx, const uint8_t *bytes, const uint32_t *offsets, const uint64_t *h,
                            uint32_t n_entries, const uint32_t *fstart, const bsg_filter_desc *desc,
                            uint32_t n_filters, uint64_t *out_words, uint64_t n_words, const SectionsOut *sections = nullptr)
{
    BSG_ENTER(ctx);
    if (!ctx) return fail(BSG_E_INVALID, "ctx is null");
    if (n_filters == 0) { if (sections) sections->sec_off[0] = 0; return BSG_OK; }
    if (!fstart || !desc || (!out_words && !sections)) return fail(BSG_E_INVALID, "null argument");
    if (int32_t rc = validate_descs(desc, n_filters, n_words)) return rc;
    if (fstart[n_filters] != n_entries) return fail(BSG_E_INVALID, "filter_entry_start[n_filters] != n_entries");
    for (uint32_t f = 0; f < n_filters; ++f)
        if (fstart[f + 1] < fstart[f]) return fail(BSG_E_INVALID, "filter_entry_start not monotone at %u", f);
    if (!h) {
        if (n_entries && !offsets) return fail(BSG_E_INVALID, "offsets is null");
        for (uint32_t e = 0; e < n_entries; ++e)
            if (offsets[e + 1] < offsets[e]) return fail(BSG_E_INVALID, "offsets not monotone at %u", e);
        if (n_entries && offsets[n_entries] && !bytes) return fail(BSG_E_INVALID, "bytes is null");
    }
    // ---- parts: contiguous runs of filters, one per device (the build shards by block; partitions are independent,
    // flush.go:191-254).  A run owns the words [first filter's offset, end of its last filter): the layout must ascend with
    // the filter index for the runs' word ranges to be disjoint — any other layout stays on one device. ----
    const uint32_t nd = (uint32_t)ctx->devs.size();
    const uint32_t unit = sections ? 3u : 1u;
    uint32_t want_parts = (nd > 1 && n_entries >= ctx->shard_min_entries) ? std::min<uint32_t>(nd, n_filters / unit) : 1;
    if (want_parts > 1) {
        uint64_t prev_end = 0;
        for (uint32_t f = 0; f < n_filters && want_parts > 1; ++f) {
            if (desc[f].m == 0) continue;
            if (desc[f].word_off < prev_end) want_parts = 1;
            prev_end = desc[f].word_off + (desc[f].m + 63) / 64;
        }
    }
    std::vector<uint32_t> cuts{0, n_filters};
    if (want_parts > 1) {
        std::vector<uint64_t> cost(n_filters);
        for (uint32_t f = 0; f < n_filters; ++f) cost[f] = (uint64_t)(fstart[f + 1] - fstart[f]) * 8 + (desc[f].m + 63) / 64 + 16;
        cuts = balanced_cuts(cost, want_parts, unit);
    }
    const uint32_t n_parts = (uint32_t)cuts.size() - 1;
    std::vector<BuildPart> parts(n_parts);
    uint64_t region_cursor = 0;
    for (uint32_t i = 0; i < n_parts; ++i) {
        BuildPart &P = parts[i];
        P.f0 = cuts[i]; P.f1 = cuts[i + 1];
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t f = P.f0; f < P.f1; ++f) {
            if (desc[f].m == 0) continue;
            lo = std::min(lo, desc[f].word_off);
            hi = std::max(hi, desc[f].word_off + (desc[f].m + 63) / 64);
        }
        if (lo == ~0ull) lo = hi = 0;
        if (n_parts == 1) { lo = 0; hi = n_words; }          // the single-device call keeps its round-1 contract: the whole arena comes back, zero-filled
        P.w_lo = lo; P.w_hi = hi;
        if (sections) {
            P.region_off = region_cursor;
            for (uint32_t b = P.f0 / 3; b < P.f1 / 3; ++b) P.region_len += section_len(desc + (size_t)b * 3);
            region_cursor += P.region_len;
        }
    }
    if (sections && region_cursor > sections->cap)
        return fail(BSG_E_INVALID, "section region needs %llu bytes, caller gave %llu", (unsigned long long)region_cursor, (unsigned long long)sections->cap);
    if (!sections && n_parts > 1) {                            // words no part owns (gaps, absent filters) are zero, as the single-device call leaves them
        uint64_t at = 0;
        for (const BuildPart &P : parts) { if (P.w_lo > at) memset(out_words + at, 0, (P.w_lo - at) * 8); at = std::max(at, P.w_hi); }
        if (n_words > at) memset(out_words + at, 0, (n_words - at) * 8);
    }
    const uint32_t first = n_parts == 1 ? pick_device(ctx) : 0;
    if (int32_t rc = run_parts(n_parts, [&](uint32_t i) -> int32_t {
            return build_on_device(ctx, *ctx->devs[(first + i) % nd], parts[i], bytes, offsets, h, fstart, desc, out_words, sections);
        })) return rc;
    float ms = 0.f, ems = 0.f;
    for (const BuildPart &P : parts) { ms = std::max(ms, P.ms); ems = std::max(ems, P.encode_ms); }
    if (sections) {
        for (const BuildPart &P : parts)
            for (uint32_t b = P.f0 / 3; b <= P.f1 / 3; ++b) sections->sec_off[b] = P.region_off + P.sec_off_local[b - P.f0 / 3];
    }
    std::lock_guard<std::shared_mutex> lk(ctx->mu);
    ctx->last_build_ms = ms;
    if (sections) ctx->last_encode_ms = ems;
    return BSG_OK;
}

int32_t bsg_build(bsg_ctx *ctx, const uint8_t *bytes, const uint32_t *offsets, uint32_t n_entries,
                  const uint32_t *filter_entry_start, const bsg_filter_desc *desc, uint32_t n_filters,
                  uint64_t *out_words, uint64_t n_words)
{
    return build_common(ctx, bytes, offsets, nullptr, n_entries, filter_entry_start, desc, n_filters, out_words, n_words);
}

int32_t bsg_build_hashed(bsg_ctx *ctx, const uint64_t *h, uint32_t n_entries, const uint32_t *filter_entry_start,
                         const bsg_filter_desc *desc, uint32_t n_filters, uint64_t *out_words, uint64_t n_words)
{
    if (n_entries && !h) return fail(BSG_E_INVALID, "h is null");
    static const uint64_t dummy = 0;
    return build_common(ctx, nullptr, nullptr, h ? h : &dummy, n_entries, filter_entry_start, desc, n_filters,
                        out_words, n_words);
}

int32_t bsg_arena_load(bsg_ctx *ctx, const uint64_t *words, uint64_t n_words, const bsg_filter_desc *desc,
                       uint32_t n_blocks, uint64_t *out_arena_id)
{
    BSG_ENTER(ctx);
    if (!ctx || !out_arena_id) return fail(BSG_E_INVALID, "null argument");
    if (n_blocks && !desc) return fail(BSG_E_INVALID, "desc is null");
    if (n_words && !words) return fail(BSG_E_INVALID, "words is null");
    if (int32_t rc = validate_descs(desc, (size_t)n_blocks * 3, n_words)) return rc;
    const uint32_t nd = (uint32_t)ctx->devs.size();
    auto arena = std::make_shared<Arena>();
    arena->n_blocks = n_blocks;
    arena->shards.resize(nd);
    // every shard is repacked and its copy ENQUEUED before any device is waited for: the devices' PCIe links run in parallel
    std::vector<std::vector<uint64_t>> images(nd);
    std::vector<std::vector<DevDesc>> descs(nd);
    hipError_t e = hipSuccess;
    for (uint32_t di = 0; di < nd && e == hipSuccess; ++di) {
        ArenaShard &s = arena->shards[di];
        Device &d = *ctx->devs[di];
        s.n_blocks = n_blocks > di ? (n_blocks - di + nd - 1) / nd : 0;
        // re-lay the shard's filters on 128-byte boundaries
        std::vector<DevDesc> &dd = descs[di];
        dd.resize((size_t)s.n_blocks * 3);
        uint64_t cursor = 0;
        for (uint32_t lb = 0; lb < s.n_blocks; ++lb) {
            const uint32_t b = lb * nd + di;
            for (uint32_t c = 0; c < 3; ++c) {
                const bsg_filter_desc &f = desc[(size_t)b * 3 + c];
                DevDesc &o = dd[(size_t)lb * 3 + c];
                o = DevDesc{0, f.m, barrett_magic(f.m), f.k, 0};
                if (f.m == 0) continue;
                const uint64_t nw = (f.m + 63) / 64;
                o.word_off = cursor;
                cursor += (nw + kAlignWords - 1) / kAlignWords * kAlignWords;
                s.sum_words[c] += nw;
                if (nw <= kLdsCapWords) s.max_staged_words[c] = std::max(s.max_staged_words[c], nw);
                if (s.fixed_m[c] == 0 && s.geometry_uniform[c]) { s.fixed_m[c] = f.m; s.fixed_k[c] = f.k; }
                else if (s.fixed_m[c] != f.m || s.fixed_k[c] != f.k) s.geometry_uniform[c] = false;
            }
        }
        s.n_words = cursor + kAlignWords;
        // one repacked host image -> one H2D copy (the caller's pointers are not retained)
        std::vector<uint64_t> &image = images[di];
        image.assign(s.n_words, 0);
        for (uint32_t lb = 0; lb < s.n_blocks; ++lb) {
            const uint32_t b = lb * nd + di;
            for (uint32_t c = 0; c < 3; ++c) {
                const bsg_filter_desc &f = desc[(size_t)b * 3 + c];
                if (f.m == 0) continue;
                memcpy(image.data() + dd[(size_t)lb * 3 + c].word_off, words + f.word_off, (f.m + 63) / 64 * 8);
            }
        }
        std::lock_guard<std::mutex> lk(d.mu);
        if (int32_t rc = use_device(d)) { free_arena(ctx, *arena); return rc; }
        e = hipMalloc(reinterpret_cast<void **>(&s.d_words), s.n_words * 8);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s.d_desc), std::max<size_t>(dd.size(), 1) * sizeof(DevDesc));
        if (e == hipSuccess && !dd.empty())
            e = hipMemcpyAsync(s.d_desc, dd.data(), dd.size() * sizeof(DevDesc), hipMemcpyHostToDevice, d.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s.d_words, image.data(), s.n_words * 8, hipMemcpyHostToDevice, d.stream);
    }
    for (uint32_t di = 0; di < nd && e == hipSuccess; ++di) {
        Device &d = *ctx->devs[di];
        std::lock_guard<std::mutex> lk(d.mu);
        (void)hipSetDevice(d.id);
        e = hipStreamSynchronize(d.stream);
    }
    if (e != hipSuccess) {
        for (auto &dp : ctx->devs) { std::lock_guard<std::mutex> lk(dp->mu); (void)hipSetDevice(dp->id); (void)hipStreamSynchronize(dp->stream); }
        free_arena(ctx, *arena);
        return fail(e == hipErrorOutOfMemory ? BSG_E_NOMEM : BSG_E_HIP, "arena upload failed: %s", hipGetErrorString(e));
    }
    std::lock_guard<std::shared_mutex> lk(ctx->mu);
    const uint64_t id = ctx->next_id++;
    ctx->arenas[id] = arena;
    *out_arena_id = id;
    return BSG_OK;
}

}  // extern "C"

namespace {
const bsg::CrcConsts &crc_consts()
{
    static bsg::CrcConsts c = [] {
        bsg::CrcConsts t{};
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t v = i;
            for (int j = 0; j < 8; ++j) v = (v & 1) ? (v >> 1) ^ bsg::kCrc32cPoly : v >> 1;
            t.table[0][i] = v;
        }
        for (int k = 1; k < 8; ++k)
            for (uint32_t i = 0; i < 256; ++i) t.table[k][i] = (t.table[k - 1][i] >> 8) ^ t.table[0][t.table[k - 1][i] & 0xFF];
        uint32_t p = 1u << 30;   // x^1
        t.x2n[0] = p;
        for (int n = 1; n < 64; ++n) t.x2n[n] = p = bsg::crc_multmodp(p, p);
        t.skip = bsg::crc_x2nmodp((uint64_t)bsg::kCrcGranule * (bsg::kDecodeThreads - 1), 3, t.x2n);
        for (uint32_t i = 0; i < 256; ++i) t.gpow[i] = bsg::crc_x2nmodp((uint64_t)bsg::kCrcGranule * i, 3, t.x2n);
        for (uint32_t i = 0; i < bsg::kCrcGranule; ++i) t.bpow[i] = bsg::crc_x2nmodp(i, 3, t.x2n);
        for (uint32_t i = 0; i < 64; ++i) t.upow[i] = bsg::crc_x2nmodp((uint64_t)BSG_DECODE_UNIT * i, 3, t.x2n);
        return t;
    }();
    return c;
}
inline uint64_t rd_be64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }
inline uint32_t rd_le32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
}  // namespace

extern "C" {

}  // extern "C"

namespace {
bool cache_owns(bsg_ctx *ctx, uint64_t arena_id);   // cache_api.inc

int32_t arena_free_by_id(bsg_ctx *ctx, uint64_t arena_id)
{
    std::shared_ptr<Arena> a;
    {
        std::lock_guard<std::shared_mutex> lk(ctx->mu);
        auto it = ctx->arenas.find(arena_id);
        if (it == ctx->arenas.end()) return fail(BSG_E_NOTFOUND, "unknown arena id %llu", (unsigned long long)arena_id);
        a = it->second;
        ctx->arenas.erase(it);
        ctx->arena_epoch.store(g_arena_epoch.fetch_add(1, std::memory_order_relaxed), std::memory_order_release);
    }
    for (auto &dp : ctx->devs) {
        std::lock_guard<std::mutex> lk(dp->mu);
        (void)hipSetDevice(dp->id);
        (void)hipStreamSynchronize(dp->stream);
    }
    free_arena(ctx, *a);
    return BSG_OK;
}
}  // namespace

extern "C" {

int32_t bsg_arena_free(bsg_ctx *ctx, uint64_t arena_id)
{
    BSG_ENTER(ctx);
    if (cache_owns(ctx, arena_id))
        return fail(BSG_E_INVALID, "arena %llu belongs to the file-arena cache (bsg_file_arena_publish): end its leases with bsg_file_arena_release", (unsigned long long)arena_id);
    return arena_free_by_id(ctx, arena_id);
}

}  // extern "C"

#include "probe_api.inc"
#include "combine_api.inc"

extern "C" {

int32_t bsg_last_kernel_ms(bsg_ctx *ctx, float *build_ms, float *hash_ms, float *decode_ms)
{
    BSG_ENTER(ctx);
    if (!ctx) return fail(BSG_E_INVALID, "ctx is null");
    std::lock_guard<std::shared_mutex> lk(ctx->mu);
    if (build_ms) *build_ms = ctx->last_build_ms;
    if (hash_ms) *hash_ms = ctx->last_hash_ms;
    if (decode_ms) *decode_ms = ctx->last_decode_ms;
    return BSG_OK;
}

// The last step of the probe as the reference takes it: evaluateBlockFilters appends blockScanCandidate{index} per surviving
// block, in the order the blocks were consulted — ascending RowDataOffset (query_exec.go:321, 603).  An arena's blocks are in
// the caller's order, so the ascending bit positions of a query's survivor row ARE that list.  Pure host arithmetic.
int32_t bsg_survivor_list(const uint64_t *survivor_row, uint32_t n_blocks, uint32_t *out_blocks, uint32_t cap, uint32_t *out_n)
{
    if (!out_n || (n_blocks && !survivor_row)) return fail(BSG_E_INVALID, "null argument");
    uint32_t n = 0;
    const uint32_t G = (n_blocks + 63) / 64;
    for (uint32_t g = 0; g < G; ++g) {
        uint64_t w = survivor_row[g];
        if (g == G - 1 && (n_blocks & 63u)) w &= (1ull << (n_blocks & 63u)) - 1;      // bits past the arena's end never count
        while (w) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(w);
            w &= w - 1;
            if (out_blocks && n < cap) out_blocks[n] = g * 64 + bit;
            ++n;
        }
    }
    *out_n = n;
    if (out_blocks && n > cap) return fail(BSG_E_INVALID, "%u blocks survive, the caller's list holds %u", n, cap);
    return BSG_OK;
}

int32_t bsg_peer_access(bsg_ctx *ctx, uint8_t *out_matrix, uint32_t n)
{
    BSG_ENTER(ctx);
    const uint32_t nd = (uint32_t)ctx->devs.size();
    if (!out_matrix || n != nd) return fail(BSG_E_INVALID, "out_matrix must hold %u x %u entries", nd, nd);
    for (uint32_t i = 0; i < nd * nd; ++i) out_matrix[i] = ctx->peer.size() == (size_t)nd * nd ? ctx->peer[i] : 0;
    return BSG_OK;
}

int32_t bsg_device_calls(bsg_ctx *ctx, uint64_t *out_calls, uint32_t cap)
{
    BSG_ENTER(ctx);
    if (!out_calls) return fail(BSG_E_INVALID, "null argument");
    for (uint32_t i = 0; i < cap && i < ctx->devs.size(); ++i) out_calls[i] = ctx->devs[i]->calls.load(std::memory_order_relaxed);
    return BSG_OK;
}

int32_t bsg_last_or_ms(bsg_ctx *ctx, float *or_ms)
{
    BSG_ENTER(ctx);
    if (!ctx || !or_ms) return fail(BSG_E_INVALID, "null argument");
    std::lock_guard<std::shared_mutex> lk(ctx->mu);
    *or_ms = ctx->last_or_ms;
    return BSG_OK;
}

int32_t bsg_timing_read(bsg_ctx *ctx, bsg_timing *out, int32_t reset)
{
    BSG_ENTER(ctx);
    if (!ctx || !out) return fail(BSG_E_INVALID, "null argument");
    for (auto &dp : ctx->devs) {
        std::lock_guard<std::mutex> lk(dp->mu);
        if (int32_t rc = use_device(*dp)) return rc;
        if (int32_t rc = drain_timing(ctx, *dp)) return rc;
    }
    std::lock_guard<std::shared_mutex> lk(ctx->mu);
    *out = ctx->timing;
    if (reset) ctx->timing = bsg_timing{};
    return BSG_OK;
}

static int32_t or_reduce_shard(bsg_ctx *ctx, Arena &arena, uint32_t di, uint32_t kind, uint64_t n_words, uint64_t *d_out)
{
    Device &d = *ctx->devs[di];
    ArenaShard &s = arena.shards[di];
    if (!s.geometry_uniform[kind]) return fail(BSG_E_INVALID, "filters of kind %u do not share (m, k)", kind);
    if (s.fixed_m[kind] != 0 && (s.fixed_m[kind] + 63) / 64 != n_words)
        return fail(BSG_E_INVALID, "n_words %llu does not match m %llu", (unsigned long long)n_words,
                    (unsigned long long)s.fixed_m[kind]);
    const uint64_t gx64 = ((n_words + 1) / 2 + bsg::kOrThreads - 1) / bsg::kOrThreads;
    if (gx64 > 0x7FFFFFFFull) return fail(BSG_E_UNSUPPORTED, "bitset of %llu words is too large for one OR launch", (unsigned long long)n_words);
    const uint32_t gx = (uint32_t)std::max<uint64_t>(gx64, 1);
    // one wave per workgroup: aim at ~8 waves per CU over the whole grid, in groups of 16..kOrMaxGroup blocks
    const uint64_t want_y = std::max<uint64_t>(1, ((uint64_t)d.n_cus * 8 + gx - 1) / gx);
    uint32_t group = (uint32_t)std::min<uint64_t>(bsg::kOrMaxGroup, std::max<uint64_t>(16, (s.n_blocks + want_y - 1) / want_y));
    if (gx >= (uint64_t)d.n_cus * 8) group = std::max(group, std::min(bsg::kOrBlocksPerGroup, bsg::kOrMaxGroup));
    if (const char *e = getenv("BSG_LAB_OR_GROUP")) group = std::min<uint32_t>(bsg::kOrMaxGroup, std::max(1, atoi(e)));   // lab only
    const uint32_t gy = std::max(1u, (s.n_blocks + group - 1) / group);
    if (gy > 65535) return fail(BSG_E_UNSUPPORTED, "%u blocks in groups of %u exceed one OR launch", s.n_blocks, group);
    if (gy > 1) HIP_TRY(hipMemsetAsync(d_out, 0, n_words * 8, d.stream));
    if (!d.kb0) { HIP_TRY(hipEventCreate(&d.kb0)); HIP_TRY(hipEventCreate(&d.kb1)); }
    hipExtLaunchKernelGGL(bsg::k_or_reduce_blocks, dim3(gx, gy), dim3(bsg::kOrThreads), 0, d.stream, d.kb0, d.kb1, 0,
                          s.d_words, s.d_desc, s.n_blocks, kind, n_words, d_out, group);
    HIP_TRY(hipGetLastError());
    d.or_pending = true;
    return BSG_OK;
}

int32_t bsg_or_reduce_dev(bsg_ctx *ctx, uint64_t arena_id, uint32_t kind, void *d_out, uint64_t n_words)
{
    BSG_ENTER(ctx);
    if (!ctx || !d_out) return fail(BSG_E_INVALID, "null argument");
    if (kind > 2) return fail(BSG_E_INVALID, "unknown kind %u", kind);
    if (ctx->devs.size() != 1) return fail(BSG_E_UNSUPPORTED, "bsg_or_reduce_dev needs a single-device context");
    std::shared_ptr<Arena> arena;
    if (int32_t rc = get_arena(ctx, arena_id, arena)) return rc;
    Device &d = *ctx->devs[0];
    std::lock_guard<std::mutex> lk(d.mu);
    if (int32_t rc = use_device(d)) return rc;
    if (int32_t rc = or_reduce_shard(ctx, *arena, 0, kind, n_words, static_cast<uint64_t *>(d_out))) return rc;
    HIP_TRY(hipStreamSynchronize(d.stream));
    if (d.or_pending) { HIP_TRY(hipEventElapsedTime(&d.last_or_ms, d.kb0, d.kb1)); d.or_pending = false; }
    { std::lock_guard<std::shared_mutex> lk2(ctx->mu); ctx->last_or_ms = d.last_or_ms; }
    return BSG_OK;
}

int32_t bsg_or_words_dev(bsg_ctx *ctx, void *d_dst, const void *d_src, uint64_t n_words, uint32_t n_src)
{
    BSG_ENTER(ctx);
    if (!ctx || !d_dst || (n_src && !d_src)) return fail(BSG_E_INVALID, "null argument");
    Device &d = *ctx->devs[0];
    std::lock_guard<std::mutex> lk(d.mu);
    if (int32_t rc = use_device(d)) return rc;
    if (n_words && n_src) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_words + 255) / 256, 4096);
        hipLaunchKernelGGL(bsg::k_or_words, dim3(grid), dim3(256), 0, d.stream, static_cast<uint64_t *>(d_dst),
                           static_cast<const uint64_t *>(d_src), n_words, n_src, 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(d.stream));
    return BSG_OK;
}

int32_t bsg_or_reduce(bsg_ctx *ctx, uint64_t arena_id, uint32_t kind, uint64_t *out_words, uint64_t n_words)
{
    BSG_ENTER(ctx);
    if (!ctx || !out_words) return fail(BSG_E_INVALID, "null argument");
    if (kind > 2) return fail(BSG_E_INVALID, "unknown kind %u", kind);
    std::shared_ptr<Arena> arena;
    if (int32_t rc = get_arena(ctx, arena_id, arena)) return rc;
    const uint32_t nd = (uint32_t)ctx->devs.size();
    // geometry must agree across shards too
    uint64_t m = 0; uint32_t k = 0;
    for (uint32_t di = 0; di < nd; ++di) {
        const ArenaShard &s = arena->shards[di];
        if (s.fixed_m[kind] == 0) continue;
        if (m == 0) { m = s.fixed_m[kind]; k = s.fixed_k[kind]; }
        else if (m != s.fixed_m[kind] || k != s.fixed_k[kind])
            return fail(BSG_E_INVALID, "filters of kind %u do not share (m, k) across devices", kind);
    }
    // Every device ORs its own shard into a partial bitset; the partials travel device-to-device (xGMI peer copies)
    // onto the first device, one k_or_words folds them, and only the result crosses PCIe.  All device locks are held
    // (taken in index order) — the reduce is a merge-time operation, not a hot concurrent one.
    std::vector<std::unique_lock<std::mutex>> locks;
    for (uint32_t di = 0; di < nd; ++di) locks.emplace_back(ctx->devs[di]->mu);
    Device &d0 = *ctx->devs[0];
    std::vector<void *> partial(nd, nullptr);
    std::vector<hipEvent_t> done(nd, nullptr);
    void *gathered = nullptr;
    auto cleanup = [&]() {
        for (uint32_t di = 0; di < nd; ++di) {
            (void)hipSetDevice(ctx->devs[di]->id);
            if (done[di]) (void)hipEventDestroy(done[di]);
            if (partial[di]) ctx->devs[di]->pool.free(partial[di]);
        }
        (void)hipSetDevice(d0.id);
        if (gathered) d0.pool.free(gathered);
    };
    int32_t rc = BSG_OK;
    for (uint32_t di = 0; di < nd && rc == BSG_OK; ++di) {
        Device &d = *ctx->devs[di];
        if ((rc = use_device(d))) break;
        hipError_t e = d.pool.alloc(&partial[di], std::max<uint64_t>(n_words, 1) * 8);
        if (e == hipSuccess && nd > 1) e = hipEventCreateWithFlags(&done[di], hipEventDisableTiming);
        if (e != hipSuccess) { rc = fail(e == hipErrorOutOfMemory ? BSG_E_NOMEM : BSG_E_HIP, "or_reduce scratch: %s", hipGetErrorString(e)); break; }
        if (arena->shards[di].n_blocks == 0) {
            if (hipMemsetAsync(partial[di], 0, n_words * 8, d.stream) != hipSuccess) { rc = fail(BSG_E_HIP, "memset failed"); break; }
        } else if ((rc = or_reduce_shard(ctx, *arena, di, kind, n_words, static_cast<uint64_t *>(partial[di])))) break;
        if (nd > 1 && hipEventRecord(done[di], d.stream) != hipSuccess) { rc = fail(BSG_E_HIP, "event record failed"); break; }
    }
    if (rc == BSG_OK) rc = use_device(d0);
    if (rc == BSG_OK && nd > 1) {
        hipError_t e = d0.pool.alloc(&gathered, (uint64_t)(nd - 1) * std::max<uint64_t>(n_words, 1) * 8);
        for (uint32_t di = 1; di < nd && e == hipSuccess; ++di) {
            e = hipStreamWaitEvent(d0.stream, done[di], 0);
            if (e == hipSuccess)
                e = peer_copy(ctx, 0, static_cast<uint64_t *>(gathered) + (uint64_t)(di - 1) * n_words, di, partial[di], n_words * 8, d0.stream);
        }
        if (e == hipSuccess && n_words) {
            const uint32_t grid = (uint32_t)std::min<uint64_t>((n_words + 255) / 256, 4096);
            hipLaunchKernelGGL(bsg::k_or_words, dim3(grid), dim3(256), 0, d0.stream, static_cast<uint64_t *>(partial[0]),
                               static_cast<const uint64_t *>(gathered), n_words, nd - 1, 0);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = fail(e == hipErrorOutOfMemory ? BSG_E_NOMEM : BSG_E_HIP, "or_reduce gather: %s", hipGetErrorString(e));
    }
    if (rc == BSG_OK) {
        hipError_t e = hipMemcpyAsync(out_words, partial[0], n_words * 8, hipMemcpyDeviceToHost, d0.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d0.stream);
        if (e != hipSuccess) rc = fail(BSG_E_HIP, "or_reduce copy out: %s", hipGetErrorString(e));
    }
    if (rc == BSG_OK) {
        float slowest = 0.f;
        for (uint32_t di = 0; di < nd; ++di) {
            Device &d = *ctx->devs[di];
            if (!d.or_pending) continue;
            (void)hipSetDevice(d.id);
            (void)hipStreamSynchronize(d.stream);
            (void)hipEventElapsedTime(&d.last_or_ms, d.kb0, d.kb1);
            d.or_pending = false;
            slowest = std::max(slowest, d.last_or_ms);
        }
        std::lock_guard<std::shared_mutex> lk2(ctx->mu);
        ctx->last_or_ms = slowest;
    } else
        for (uint32_t di = 0; di < nd; ++di) { (void)hipSetDevice(ctx->devs[di]->id); (void)hipStreamSynchronize(ctx->devs[di]->stream); }
    cleanup();
    return rc;
}

}  // extern "C"

#include "encode_api.inc"
#include "ingest_api.inc"
#include "match_api.inc"
#include "comm_api.inc"
#include "stream_api.inc"
#include "cache_api.inc"
