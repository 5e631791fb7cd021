/*
 * bloomgpu.h — C-ABI of libbloomgpu.so: bloomsearch's hierarchical bloom-filter
 * construct + probe hot path on AMD MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary.  The reference (github.com/danthegoodman1/bloomsearch,
 * pure Go) has no plugin seam at the bloom level, so the Go host binds these entry
 * points with cgo at exactly the internal seams named beside each function
 * (see INTEGRATION.md for the cgo stub).  Everything here is plain C99: int32
 * status returns, opaque context, caller-owned pointers that are only read or
 * written for the duration of the call (cgo rule: C never retains a Go pointer).
 *
 * Semantics replaced (reference file:line):
 *   bsg_hash_entries / bsg_build ... buildSizedBloomFilter's AddString loop, ingest.go:139-145
 *                                    (bloom/v3 v3.7.0 baseHashes + location + bitset.Set)
 *   bsg_arena_load ................. the decoded result of blockFilterCursor.filtersFor /
 *                                    parseFilterSection for every block, file_format.go:392-448,575
 *   bsg_batch_create ............... the pruneBloomQuery tree, query_exec.go:220 / query.go:586-718,
 *                                    with each distinct term hashed ONCE instead of per TestString
 *   bsg_probe / bsg_probe_batch .... evaluateBlockFilters' per-block loop + evaluateBloomFilters /
 *                                    evaluateBloomExpression / evaluateBloomCondition,
 *                                    query_exec.go:572-615 and :75-159
 *   bsg_or_reduce .................. north-star extension (fixed-geometry OR of block filters into a
 *                                    file-level filter); the reference rebuilds instead, merge.go:447-453
 *
 * Threading: every entry point is re-entrant on one context (internal mutex
 * around handle tables, per-call stream ordering); no thread-local "current
 * device" is assumed.  No callbacks, no exceptions cross the boundary.
 */
#ifndef BLOOMGPU_H
#define BLOOMGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSG_API __attribute__((visibility("default")))

/* ---- status codes ---- */
#define BSG_OK              0
#define BSG_E_INVALID      -1  /* bad argument / malformed descriptor or program (rejected before any launch) */
#define BSG_E_HIP          -2  /* HIP runtime error (message in bsg_last_error) */
#define BSG_E_NOMEM        -3
#define BSG_E_NOTFOUND     -4  /* unknown arena / batch id */
#define BSG_E_UNSUPPORTED  -5
#define BSG_E_NODEVICE     -6  /* no gfx950 device visible: the library never falls back to a CPU path */

/* ---- filter kinds: which of a block's three filters a term tests
 *      (BloomField / BloomToken / BloomFieldToken, query.go:480-484) ---- */
#define BSG_KIND_FIELD        0u
#define BSG_KIND_TOKEN        1u
#define BSG_KIND_FIELD_TOKEN  2u

/* ---- query program opcodes: op = (opcode << 28) | arg, postfix order ----
 * TERM i : push TestString(term i) against the block's filter of term i's kind;
 *          absent filter => true (fail-open, query_exec.go:137-151)
 * AND n  : pop n, push conjunction   (n == 0 => true,  query_exec.go:115-121)
 * OR  n  : pop n, push disjunction   (n == 0 => false, query_exec.go:105-114)
 * TRUE   : nil expression / nil condition           (query_exec.go:84,97)
 * FALSE  : unknown expression or condition type     (query_exec.go:122,156)
 * A query with zero ops is a nil BloomQuery => true (query_exec.go:81-83). */
#define BSG_OP_TERM   0u
#define BSG_OP_AND    1u
#define BSG_OP_OR     2u
#define BSG_OP_TRUE   3u
#define BSG_OP_FALSE  4u
#define BSG_OP(opcode, arg) (((uint32_t)(opcode) << 28) | ((uint32_t)(arg) & 0x0FFFFFFFu))

/* One distinct query term: the four bloom/v3 base hashes of the probed string
 * (Field, Token, or Field+"::"+Token — tokenizer.go:509-511) and the filter kind. */
typedef struct bsg_term {
    uint64_t h[4];
    uint32_t kind;
    uint32_t reserved;
} bsg_term;

/* One filter inside a word arena.  Bit i of the filter is bit (i & 63) of
 * words[word_off + (i >> 6)] (bitset v1.10.0 layout; native little-endian u64,
 * i.e. the Go []uint64 as-is — the big-endian swap of WriteTo/ReadFrom is the
 * wire format only).  m == 0 marks an absent (nil) filter. */
typedef struct bsg_filter_desc {
    uint64_t word_off;
    uint64_t m;
    uint32_t k;
    uint32_t reserved;
} bsg_filter_desc;

/* Device-side timing of probes issued with BSG_PROBE_TIMED (HIP events on the
 * library's own stream; accumulated until read). */
typedef struct bsg_timing {
    uint64_t n_probes;           /* timestamped k_probe_terms dispatches                                    */
    double   ms_terms_kernel;    /* sum of their durations (the HBM stream)                                 */
    double   ms_eval_kernel;     /* sum of the timestamped k_eval_programs durations (n_eval of them)       */
    uint64_t stream_bytes;       /* bitset bytes streamed by those k_probe_terms dispatches                 */
    uint64_t n_probe_arenas;     /* arenas those dispatches covered (one dispatch probes a group of arenas) */
    uint64_t n_eval;
    uint64_t n_fused;            /* timestamped k_probe_fused dispatches (probe of group i + eval of i-1)   */
    double   ms_fused_kernel;
    uint64_t fused_stream_bytes; /* bitset bytes streamed by their probe role                               */
    uint64_t n_fused_arenas;
    uint64_t n_folded;           /* timestamped k_probe_eval dispatches (probe + per-tile program evaluation in one) */
    double   ms_folded_kernel;
    uint64_t folded_stream_bytes;/* bitset bytes streamed by them                                           */
    uint64_t n_folded_arenas;
} bsg_timing;

#define BSG_PROBE_ASYNC  1u  /* enqueue only; caller later calls bsg_sync                    */
#define BSG_PROBE_TIMED  2u  /* timestamp the dispatches (see bsg_timing_read)               */
#define BSG_PROBE_NOFUSE 4u  /* never fuse or fold: every group runs as k_probe_terms + k_eval_programs */
#define BSG_PROBE_ROWS_PACKED 8u /* bsg_probe_many_rows only: LIST / DENSE payloads packed per run of 256 queries (below)    */

typedef struct bsg_ctx bsg_ctx;

/* ---- lifecycle (engine construct / Stop) ---- */
BSG_API int32_t bsg_device_count(void);
/* out_matrix[i * n + j] (n = the context's devices) = 1 when device i reaches device j's memory directly (peer access over xGMI
 * enabled at bsg_open, or the same device), 0 when copies between the two are staged through host memory by the runtime:
 * still correct, at PCIe speed, and announced on stderr once per pair (BSG_QUIET=1 silences it). */
BSG_API int32_t bsg_peer_access(bsg_ctx *ctx, uint8_t *out_matrix, uint32_t n);
BSG_API int32_t bsg_open(const int32_t *device_ids, int32_t n_devices, bsg_ctx **out_ctx);
BSG_API int32_t bsg_close(bsg_ctx *ctx);
/* ---- errors ----
 * Every failing call records its message on the handle it was made on; bsg_last_error(handle) returns a private copy of
 * it (valid until the calling thread's next bsg_last_error) and may run on ANY OS thread — nothing here depends on the
 * caller staying on one thread between the failing call and the read (a goroutine may migrate between two cgo calls,
 * SURVEY 7 hard part 6: no runtime.LockOSThread).  A context shared by concurrent callers has ONE slot, so a Go host
 * gives every goroutine-confined caller (flush worker, merge, each query's file worker) a SCOPE of its own:
 * bsg_scope_open returns a lightweight alias of the context — valid wherever a bsg_ctx* is — that owns nothing but its
 * own error slot; close it with bsg_close before the context.  bsg_last_error(NULL) serves only the context-free entry
 * points (bsg_open, bsg_estimate_parameters, bsg_sections_size) and is thread-local; bsg_open_err returns bsg_open's
 * message in the same call instead. */
BSG_API const char *bsg_last_error(bsg_ctx *ctx);
BSG_API int32_t bsg_last_error_copy(bsg_ctx *ctx, char *buf, uint64_t cap);
BSG_API int32_t bsg_scope_open(bsg_ctx *ctx, bsg_ctx **out_scope);
BSG_API int32_t bsg_open_err(const int32_t *device_ids, int32_t n_devices, bsg_ctx **out_ctx, char *errbuf, uint64_t cap);
BSG_API int32_t bsg_sync(bsg_ctx *ctx);

/* Construct / match work on a context over several devices (bsg_hash_entries, bsg_build*, bsg_ingest_*, bsg_match_rows): a call
 * large enough is cut into one part per device — contiguous runs of entries / filters / sets / rows, each on a thread of its
 * own (partitions are independent in flush.go:191-254, surviving blocks in query_exec.go:729-764) — and a smaller call takes ONE
 * device, chosen round-robin among those nobody holds, so the flush worker, a merge and concurrent queries' block workers
 * land on different GPUs.  Results do not depend on the number of devices.  out_calls[i]: parts device i has served so far. */
BSG_API int32_t bsg_device_calls(bsg_ctx *ctx, uint64_t *out_calls, uint32_t cap);

/* ---- sizing: bloom/v3 EstimateParameters + New's clamps, as called by
 * buildSizedBloomFilter with n = max(len(set), 1) (ingest.go:139-140).  Pure host
 * arithmetic for non-Go hosts; a Go host keeps calling bloom.EstimateParameters so
 * libm differences can never change (m, k). ---- */
BSG_API int32_t bsg_estimate_parameters(uint64_t n, double false_positive_rate, uint64_t *m, uint64_t *k);

/* ---- construct ---- */

/* h[e] = bloom/v3 baseHashes(entry e): (murmur3_x64_128(d), murmur3_x64_128(d || 0x01)), seed 0.
 * entry e = bytes[offsets[e] .. offsets[e+1]).  out_h: n_entries * 4 u64. */
BSG_API int32_t bsg_hash_entries(bsg_ctx *ctx, const uint8_t *bytes, const uint32_t *offsets,
                                 uint32_t n_entries, uint64_t *out_h);

/* Build n_filters bitsets in one pass.  Entries are grouped by filter:
 * filter f owns entries [filter_entry_start[f], filter_entry_start[f+1]).
 * desc[f] gives (m, k) (computed by the caller, see above) and word_off into
 * out_words, which the callee zero-fills and then populates: for every entry
 * and i < k, bit (location(h, i) mod m) is set.  Result is a pure function of
 * (entry set, m, k) — insertion order is irrelevant, exactly as for AddString. */
BSG_API int32_t bsg_build(bsg_ctx *ctx, const uint8_t *bytes, const uint32_t *offsets, uint32_t n_entries,
                          const uint32_t *filter_entry_start, const bsg_filter_desc *desc, uint32_t n_filters,
                          uint64_t *out_words, uint64_t n_words);

/* Same, from pre-computed base hashes (n_entries * 4 u64). */
BSG_API int32_t bsg_build_hashed(bsg_ctx *ctx, const uint64_t *h, uint32_t n_entries,
                                 const uint32_t *filter_entry_start, const bsg_filter_desc *desc,
                                 uint32_t n_filters, uint64_t *out_words, uint64_t n_words);

/* ---- probe ---- */

/* Upload the filters of n_blocks blocks: desc[b*3 + kind].  Blocks are sharded
 * round-robin over the context's devices (block b -> device b % n_devices). */
BSG_API int32_t bsg_arena_load(bsg_ctx *ctx, const uint64_t *words, uint64_t n_words,
                               const bsg_filter_desc *desc, uint32_t n_blocks, uint64_t *out_arena_id);
/* Same, straight from the on-disk bytes: section b = region[sec_off[b] .. sec_off[b+1]) exactly as
 * encodeFilterSection wrote it (file_format.go:343-384; an empty range = block without filters).
 * Everything inside a section — CRC32C, flags, lengths, (m, k), the big-endian words — is parsed and decoded on the
 * device; the host never looks inside.  out_status[b]: 0 ok, else parseFilterSection's failure for that block (-1 too
 * small, -2 CRC mismatch = ErrInvalidHash, -3 unknown flags, -4 truncated, -5 bad filter, -6 trailing bytes, -7 the
 * section's bytes were never completely handed over); a failed block gets nil filters and never poisons the others
 * (query_exec.go:580-590).  Deviations from bloom/v3 ReadFrom, both "bad filter": a bitset shorter than m, and
 * k > 1024 (a corrupt section with a valid CRC must not make a probe loop 2^32 times).  Contexts on several devices
 * shard the blocks round-robin as bsg_arena_load does. */
BSG_API int32_t bsg_arena_load_sections(bsg_ctx *ctx, const uint8_t *region, uint64_t region_len, const uint64_t *sec_off,
                                        uint32_t n_blocks, int32_t *out_status, uint64_t *out_arena_id);
/* The same load as the reference's region cursor performs it (blockFilterCursor / planBlockFilterReads,
 * file_format.go:511-662): the host hands the file's filter region over in the chunks it reads (<= 4 MiB each in the
 * reference), in any order, skipping what it does not need; a section is decoded as soon as its last byte has
 * arrived, while the next chunk is still being copied.
 *   begin : sec_begin[b] / sec_end[b] = FILE offsets of candidate block b's section (equal = no section)
 *   append: bytes[0 .. len) = file bytes [file_offset, file_offset + len); copied out before the call returns.  Ranges that
 *           were handed over before are ignored: a byte's FIRST delivery is final (a re-read after a bad read must go
 *           through a new stream)
 *   finish: out_status[n_blocks] as above, the arena id; the stream id is consumed (abort discards it instead) */
BSG_API int32_t bsg_arena_stream_begin(bsg_ctx *ctx, const uint64_t *sec_begin, const uint64_t *sec_end, uint32_t n_blocks,
                                       uint64_t *out_stream_id);
BSG_API int32_t bsg_arena_stream_append(bsg_ctx *ctx, uint64_t stream_id, uint64_t file_offset, const uint8_t *bytes, uint64_t len);
BSG_API int32_t bsg_arena_stream_finish(bsg_ctx *ctx, uint64_t stream_id, int32_t *out_status, uint64_t *out_arena_id);
BSG_API int32_t bsg_arena_stream_abort(bsg_ctx *ctx, uint64_t stream_id);
BSG_API int32_t bsg_arena_free(bsg_ctx *ctx, uint64_t arena_id);

/* ---- resident file arenas across queries ----
 * The reference re-reads and re-parses a file's block-filter sections on every query (blockFilterCursor, file_format.go:511-662;
 * evaluateBlockFilters, query_exec.go:546-615).  Here a file's decoded filters stay on the device between queries, under one
 * policy inside the library: least recently used by BYTES against a budget, an arena somebody holds a lease on is never evicted,
 * only clean decodes become resident, a miss widens the resident arena to the union of the block sets, a tombstoned file
 * (merge.go:178-185) is forgotten at once and freed by its last user.
 *   key            the file's identity (MaybeFile.Pointer bytes)
 *   block_keys     strictly ascending u64 per candidate block (DataBlockMetadata.RowDataOffset, the order evaluateBlockFilters
 *                  walks them in); row i of a published arena holds block_keys[i]
 *   acquire        covered: *out_lease != 0, *out_arena_id, *out_arena_blocks (may be NULL) = blocks the arena holds (its survivor
 *                  rows have ceil(that / 64) words), out_rows[i] = candidate i's row (block index) in that arena;
 *                  not covered: *out_lease == 0 — load (have + own candidates), publish
 *   have           cap == 0: *out_n only; else the resident arena's block keys and section extents (file offsets)
 *   publish        arena_id from bsg_arena_stream_finish / bsg_arena_load_sections of exactly these blocks, status = its out_status
 *                  (NULL: all clean).  The cache OWNS the arena from here on (bsg_arena_free on it is BSG_E_INVALID); the returned
 *                  lease keeps it alive for this query whether or not it became resident (*out_resident).
 *   release        ends a lease (unknown / already released: BSG_E_NOTFOUND)
 * Thread-safe; device memory is freed outside the cache's lock. */
typedef struct bsg_arena_cache_stats {
    uint64_t budget_bytes, resident_bytes, resident_files, leases;
    uint64_t leased_dead_bytes;      /* arenas alive only through leases (evicted-from-table, dirty, narrower, forgotten while in use) */
    uint64_t hits, misses, published, widenings, evictions, forgotten;
    uint64_t rejected_dirty, rejected_over_budget, rejected_narrower;
} bsg_arena_cache_stats;
BSG_API int32_t bsg_set_arena_budget(bsg_ctx *ctx, uint64_t bytes);
BSG_API int32_t bsg_file_arena_acquire(bsg_ctx *ctx, const uint8_t *key, uint32_t key_len, const uint64_t *block_keys, uint32_t n_blocks,
                                       uint64_t *out_lease, uint64_t *out_arena_id, uint32_t *out_arena_blocks, uint32_t *out_rows);
BSG_API int32_t bsg_file_arena_have(bsg_ctx *ctx, const uint8_t *key, uint32_t key_len, uint64_t *out_block_keys, uint64_t *out_sec_begin,
                                    uint64_t *out_sec_end, uint32_t cap, uint32_t *out_n);
BSG_API int32_t bsg_file_arena_publish(bsg_ctx *ctx, const uint8_t *key, uint32_t key_len, uint64_t arena_id, const uint64_t *block_keys,
                                       const uint64_t *sec_begin, const uint64_t *sec_end, const int32_t *status, uint32_t n_blocks,
                                       uint64_t *out_lease, int32_t *out_resident);
BSG_API int32_t bsg_file_arena_release(bsg_ctx *ctx, uint64_t lease);
BSG_API int32_t bsg_file_arena_forget(bsg_ctx *ctx, const uint8_t *key, uint32_t key_len);
BSG_API int32_t bsg_arena_cache_stats_read(bsg_ctx *ctx, bsg_arena_cache_stats *out, int32_t reset);

/* Compile + upload a batch of queries: n_terms distinct terms and, per query q,
 * the postfix program prog_ops[prog_off[q] .. prog_off[q+1]).  No limit on the batch: one launch holds ~22 000 distinct terms
 * of a kind and ~100 verdict words per 256-query chunk, and a batch beyond that is cut — inside the library — into runs of
 * queries that probe one after the other (the caller sees one batch and one result; evaluateBloomExpression has no such limit,
 * query_exec.go:89-126).  Only a SINGLE query beyond those limits is BSG_E_UNSUPPORTED, and such a batch cannot leave its
 * survivors at a device pointer (bsg_probe_many_dev). */
BSG_API int32_t bsg_batch_create(bsg_ctx *ctx, const bsg_term *terms, uint32_t n_terms,
                                 const uint32_t *prog_ops, const uint32_t *prog_off, uint32_t n_queries,
                                 uint64_t *out_batch_id);
BSG_API int32_t bsg_batch_free(bsg_ctx *ctx, uint64_t batch_id);

/* Evaluate every query of the batch against every block of the arena.
 * out_survivors[q * ceil(n_blocks/64) + (b >> 6)] bit (b & 63) == 1  <=>  block b
 * survives query q (evaluateBloomFilters returned true).  out_survivors may be
 * NULL to leave the result on the device (benchmarks). */
BSG_API int32_t bsg_probe_batch(bsg_ctx *ctx, uint64_t arena_id, uint64_t batch_id, uint32_t flags,
                                uint64_t *out_survivors);

/* Probe the same batch against each of n_arenas arenas (e.g. the candidate files of one query
 * stage) in one call.  Up to 1 024 arenas (bsg_set_probe_group; at most 4 096) are covered by ONE dispatch — a 35 MB arena streams in
 * about the time a dispatch takes to ramp up and complete, so per-arena launches cap the HBM roofline fraction near
 * one half — and dispatches are software-pipelined: the program evaluation of group i rides inside the launch that
 * streams group i+1's bitsets (k_probe_fused).  out_survivors == NULL: enqueue only (results stay on the device; pair
 * with bsg_sync).  Otherwise the call is synchronous (unless BSG_PROBE_ASYNC is set on a single-device context: then
 * out_survivors must be C memory that stays valid until the next bsg_sync) and arena i's survivors ([n_queries][ceil(n_blocks_i / 64)] u64)
 * are written back to back in arena order; they leave the device on a copy stream while the next group is being
 * probed (pass memory from bsg_pinned_alloc for a plain DMA).  Contexts opened on several devices probe their shards
 * concurrently and interleave the shards' bitsets on the host. */
BSG_API int32_t bsg_probe_many(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas, uint64_t batch_id,
                               uint32_t flags, uint64_t *out_survivors);
/* Same, survivors left at a DEVICE pointer (single-device contexts; one-process-per-GPU layers that forward them). */
BSG_API int32_t bsg_probe_many_dev(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas, uint64_t batch_id,
                                   uint32_t flags, void *d_out_survivors);
/* Arenas one probe dispatch may cover (1..4096; 0 = the default, 1 024).  Up to 128 arena records ride in the dispatch's kernel
 * arguments; a larger group uploads its records into device memory in front of the dispatch (same stream). */
BSG_API int32_t bsg_set_probe_group(bsg_ctx *ctx, uint32_t max_arenas_per_launch);

/* ---- survivor ROWS: surviving block ids for the host, not Q x B / 8 bytes whatever they hold ----
 * bsg_probe_many with the host-side gather of the north star in mind (query_exec.go:321,603: the consumer walks the surviving
 * block INDICES of a query).  Per (arena i, query q) a header out_hdr[i * n_queries + q] = tag << 30 | surviving-block count and
 * the row's slot out_rows[row offset as in bsg_probe_many] whose content depends on the tag:
 *   BSG_ROW_NONE   no block survives: slot untouched          BSG_ROW_ALL    every block survives: slot untouched
 *   BSG_ROW_LIST   count <= 2 * ceil(n_blocks / 64): the slot starts with `count` ascending block indices (u32)
 *   BSG_ROW_DENSE  the slot holds the row's words exactly as bsg_probe_many writes them
 * Both buffers are written BY THE DEVICE (k_survivor_rows) and must be page-locked C memory (bsg_pinned_alloc /
 * bsg_host_register): only the bytes written cross PCIe — a batch whose rows are mostly NONE / ALL / short lists costs 4 bytes
 * per row instead of n_blocks / 8.  Flags as bsg_probe_many (with BSG_PROBE_ASYNC both buffers must stay valid
 * until bsg_sync).  bsg_survivor_row_list expands one row to its ascending block indices whatever its tag.
 * BSG_PROBE_ROWS_PACKED (round 6): a store into host memory that does not fill a line is one PCIe write of its own, and the dense
 * layout makes one per LIST row.  With the flag the payloads of every run of 256 consecutive queries (q / 256) of an arena lie back
 * to back, in query order, from the start of the run's slot area out_rows[row offset of the run's first query]: a LIST row takes
 * ceil(count / 2) words (its ids, u32), a DENSE row ceil(n_blocks / 64) words, NONE / ALL rows nothing — a row's payload begins
 * where the payloads of the run's earlier rows end, which the run's headers tell.  The headers shrink too: ONE BYTE per row at byte
 * (i * n_queries + q) of out_hdr (device d's slice: from byte d * n_arenas * n_queries), tag << 6 | count of a LIST row (at most
 * 32); an ALL row counts the arena's blocks, a DENSE row the bits of its words.  Buffer sizes as without the flag (the packed form
 * uses the front of both); arenas of at most 1 024 blocks per device (BSG_E_UNSUPPORTED beyond, before anything is launched).
 * bsg_survivor_rows_list_packed reads such rows. */
#define BSG_ROW_NONE  0u
#define BSG_ROW_ALL   1u
#define BSG_ROW_LIST  2u
#define BSG_ROW_DENSE 3u
BSG_API int32_t bsg_probe_many_rows(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas, uint64_t batch_id, uint32_t flags,
                                    uint64_t *out_rows, uint32_t *out_hdr);
BSG_API int32_t bsg_survivor_row_list(uint32_t hdr, const uint64_t *row, uint32_t n_blocks, uint32_t *out_blocks, uint32_t cap,
                                      uint32_t *out_n);
/* On a context of nd > 1 devices — the Go host's shape: ONE process that opens all 8 GPUs — every device writes the rows of ITS shards
 * (local block numbers: local l on device d is global block l * nd + d) into a slice of its own of the same two buffers:
 *   headers: out_hdr + d * n_arenas * n_queries, [arena][query] (count = the shard's surviving blocks)
 *   rows   : device d's slice behind the slices of the devices before it; in it arena i's [n_queries][ceil(local blocks / 64)] slots
 * bsg_survivor_rows_size gives the words / headers the two buffers must hold (nd == 1: the layout described above), and
 * bsg_survivor_rows_list — the north star's "host-side gather of surviving block IDs" (query_exec.go:603) — merges the nd shards'
 * rows of one (arena, query) into the ascending GLOBAL block indices: NONE shards contribute nothing, ALL shards every block they
 * hold, neither touches the rows' payload.  (Single-row, single-device: bsg_survivor_row_list above.) */
BSG_API int32_t bsg_survivor_rows_size(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas, uint64_t batch_id,
                                       uint64_t *out_row_words, uint64_t *out_hdr_words);
BSG_API int32_t bsg_survivor_rows_list(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas, uint64_t batch_id, const uint64_t *rows,
                                       const uint32_t *hdr, uint32_t arena_index, uint32_t query, uint32_t *out_blocks, uint32_t cap,
                                       uint32_t *out_n);
BSG_API int32_t bsg_survivor_rows_list_packed(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas, uint64_t batch_id, const uint64_t *rows,
                                              const uint32_t *hdr, uint32_t arena_index, uint32_t query, uint32_t *out_blocks, uint32_t cap,
                                              uint32_t *out_n);

/* One-shot convenience: batch_create + probe_batch + batch_free. */
BSG_API int32_t bsg_probe(bsg_ctx *ctx, uint64_t arena_id, const bsg_term *terms, uint32_t n_terms,
                          const uint32_t *prog_ops, const uint32_t *prog_off, uint32_t n_queries,
                          uint64_t *out_survivors);

/* One interactive Query() in ONE call — strings in, survivors out (Query -> evaluateBlockFilters, query_exec.go:201-225, 572-615).
 * Term t = the probed STRING term_bytes[term_off[t] .. term_off[t + 1]) (Field, Token, or Field + "::" + Token) and its filter
 * kind; programs as for bsg_batch_create; out_survivors as for bsg_probe_many (arena i's [n_queries][ceil(n_blocks_i / 64)] words
 * back to back).  The strings are hashed on the host inside the call (TestString hashes inside the call too — a device launch for
 * three strings costs more than the probe), and for <= 16 distinct terms, <= 256 queries whose lowered programs hold <= 128
 * words in all, and <= 32 arenas per device, the hashes and programs ride in the kernel arguments of ONE dispatch per device
 * (k_query_direct): no batch object, nothing uploaded, survivors written straight into page-locked memory with a doorbell.
 * Anything larger takes the batch path inside the same call.  Re-entrant like every other entry point. */
BSG_API int32_t bsg_query(bsg_ctx *ctx, const uint64_t *arena_ids, uint32_t n_arenas,
                          const uint8_t *term_bytes, const uint32_t *term_off, const uint32_t *term_kinds, uint32_t n_terms,
                          const uint32_t *prog_ops, const uint32_t *prog_off, uint32_t n_queries, uint64_t *out_survivors);

/* Concurrent bsg_query calls on one context SHARE dispatches (the reference consults every candidate file of a Query() from a worker
 * goroutine of its own, query_exec.go:303-357, 427-431, and runs several Query() calls at once — mirrored call by call that is one
 * ~8 us dispatch per (query, file), serialised on the device's stream).  A call that finds the device idle goes alone, at once, as
 * described above.  Calls that arrive while another is collecting or in flight queue inside the library; the head of the queue
 * collects everything queued as (call, arena) pairs: an arena asked enough in the cycle (8 three-term queries per 35 KB of filters per block) is STREAMED once for
 * all of them (their query sets merged into one batch, each distinct term probed once: k_probe_terms + k_eval_programs), every other
 * pair is a job of ONE k_query_jobs dispatch (gather regime: cost follows the pairs asked for — one query, one call per candidate
 * file is a list of such jobs).  The collector hands its role on (two cycles in flight) and deals every caller its rows.  Nobody
 * waits for a window to fill; results equal the solo path's bit for bit.  Calls beyond 16 terms / 128 program words / 64 queries /
 * 32 arenas always go alone (the combiner's lab switches: bloomgpu_lab.h).  bsg_query_stats_read: how calls were served. */
typedef struct bsg_query_stats {
    uint64_t calls;                /* bsg_query calls that were eligible for combining                      */
    uint64_t solo_calls;           /* ... served by a dispatch of their own                                */
    uint64_t cycles;               /* collector cycles (a solo call is a cycle of one)                      */
    uint64_t cycle_calls;          /* calls served by those cycles                                          */
    uint64_t dispatches;           /* dispatches (per device) the combined cycles enqueued: hot arenas + job lists */
    uint64_t hot_arenas;           /* ... of which arenas streamed once for all their callers               */
    uint64_t max_calls_per_cycle;
    /* the combined cycles' phases on their collectors' clocks, summed (ns): merging + planning the batches; enqueueing (tables up,
     * dispatches); waiting for the device; dealing the rows out + releasing the callers (ns_wake: the releasing part of that) */
    uint64_t ns_prepare;
    uint64_t ns_enqueue;
    uint64_t ns_wait;
    uint64_t ns_deal;
    uint64_t ns_wake;
    uint64_t ns_scatter;           /* parts of ns_deal: rows copied out ...                                */
    uint64_t ns_free;              /* ... scratch given back ...                                            */
    uint64_t ns_retire;            /* ... slot released + next collector appointed                          */
} bsg_query_stats;
BSG_API int32_t bsg_query_stats_read(bsg_ctx *ctx, bsg_query_stats *out, int32_t reset);

/* The surviving blocks of ONE query as the reference's probe hands them on: blockScanCandidate{index} per survivor in the order
 * the blocks were consulted (ascending RowDataOffset, query_exec.go:321, 603) = the ascending bit positions of the query's row
 * survivor_row[ceil(n_blocks / 64)] of a bsg_probe* / bsg_query result, for an arena loaded in that block order.  out_blocks may be
 * NULL to ask for the count.  Host arithmetic, no context. */
BSG_API int32_t bsg_survivor_list(const uint64_t *survivor_row, uint32_t n_blocks, uint32_t *out_blocks, uint32_t cap, uint32_t *out_n);

BSG_API int32_t bsg_timing_read(bsg_ctx *ctx, bsg_timing *out, int32_t reset);
/* Device time (the dispatch's own start/stop timestamps) of the most recent k_build / k_hash_entries /
 * k_decode_sections launch made through bsg_build* / bsg_hash_entries / bsg_arena_load_sections (the slowest device
 * of the call, when it was cut over several; any pointer may be NULL). */
BSG_API int32_t bsg_last_kernel_ms(bsg_ctx *ctx, float *build_ms, float *hash_ms, float *decode_ms);

/* ---- fixed-geometry OR-reduce (extension) ----
 * All present filters of `kind` in the arena must share (m, k).  out_words
 * (ceil(m/64) u64) receives their bitwise OR == build(union of entry sets, m, k). */
BSG_API int32_t bsg_or_reduce(bsg_ctx *ctx, uint64_t arena_id, uint32_t kind, uint64_t *out_words, uint64_t n_words);

/* Device-pointer helper for the one-process-per-GPU layer (torch.distributed all_gather of
 * partial bitsets, then a local OR): d_dst[i] |= d_src[s*n_words + i] for s < n_src.
 * Pointers are device pointers on the context's first device. */
BSG_API int32_t bsg_or_words_dev(bsg_ctx *ctx, void *d_dst, const void *d_src, uint64_t n_words, uint32_t n_src);
/* Per-device partial OR left on the device: writes n_words u64 at d_out. */
BSG_API int32_t bsg_or_reduce_dev(bsg_ctx *ctx, uint64_t arena_id, uint32_t kind, void *d_out, uint64_t n_words);
/* Device time of the most recent k_or_reduce_blocks dispatch (the slowest device's). */
BSG_API int32_t bsg_last_or_ms(bsg_ctx *ctx, float *or_ms);

/* ---- the OR-reduce across GPUs: RCCL over xGMI inside the library (a Go host cannot call torch.distributed) ----
 * RCCL has no bitwise-OR reduction: the all-reduce is reduce-scatter + all-gather with the OR done by a kernel of the library —
 * slice j of every rank's partial bitset travels to rank j (grouped ncclSend / ncclRecv, one slice per point-to-point link),
 * is OR-ed there, and the reduced slices are ncclAllGather-ed: 2 (world - 1) / world of the bitset per GPU on the wire.  librccl is bound
 * at run time; without it these calls fail with BSG_E_UNSUPPORTED and everything else keeps working.  (BSG_RCCL_LIBRARY in the
 * environment names the library the communicator symbols are bound from instead: an RCCL build under test, or the suite's
 * in-process loopback, tests/loopback_ccl.cpp, through which one GPU runs the world > 1 schedule.)
 *   one process per GPU : rank 0 calls bsg_comm_unique_id and hands the 128 bytes to the other ranks (any channel);
 *                         every rank calls bsg_comm_init(ctx, id, rank, world) on its single-device context.
 *   one process, N GPUs : bsg_comm_init(ctx, NULL, 0, 0) makes every device of the context a rank (ncclCommInitAll).
 * bsg_or_allreduce = bsg_or_reduce over the whole communicator: every rank's out_words receives the same bitset. */
#define BSG_COMM_ID_BYTES 128
BSG_API int32_t bsg_comm_unique_id(uint8_t *out_id);
BSG_API int32_t bsg_comm_init(bsg_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world);
BSG_API int32_t bsg_comm_destroy(bsg_ctx *ctx);
/* What the communicator library itself reports for the context's (first) communicator — ncclCommCount / ncclCommUserRank:
 * the ranks RCCL sees, not the numbers the caller passed to bsg_comm_init.  *from_library = 0 when the bound library lacks
 * the two symbols (the values of bsg_comm_init are returned then).  Any out pointer may be NULL. */
BSG_API int32_t bsg_comm_info(bsg_ctx *ctx, int32_t *out_world, int32_t *out_rank, int32_t *from_library);
BSG_API int32_t bsg_or_allreduce(bsg_ctx *ctx, uint64_t arena_id, uint32_t kind, uint64_t *out_words, uint64_t n_words);
/* In place on device memory: d_words[i] = n_words u64 on the context's device i (one pointer for a single-device context). */
BSG_API int32_t bsg_or_allreduce_dev(bsg_ctx *ctx, void *const *d_words, uint64_t n_words);

/* ---- filter sections written on the device (encodeFilterSection, file_format.go:343-384) ----
 * Section b = the three filters desc[3b .. 3b+2] (m == 0 => absent, flag bit clear):
 *   [u8 flags] { [u32 LE 24 + 8 nw] [u64 BE m] [u64 BE k] [u64 BE m] [nw x u64 BE words] }* [u32 LE CRC32C of all before]
 * byte for byte what bloom/v3 WriteTo + the reference's framing produce, so the host appends the region to the file
 * as it is.  bsg_sections_size tells how large out_region must be (a function of the geometry alone). */
BSG_API int32_t bsg_sections_size(const bsg_filter_desc *desc, uint32_t n_blocks, uint64_t *out_total);
/* bsg_build followed by the encode, without the words crossing PCIe: n_filters = 3 * n_blocks; n_words sizes the
 * device arena the descriptors' word_off address; out_sec_off[n_blocks + 1] receives each section's byte offset. */
BSG_API int32_t bsg_build_sections(bsg_ctx *ctx, const uint8_t *bytes, const uint32_t *offsets, uint32_t n_entries,
                                   const uint32_t *filter_entry_start, const bsg_filter_desc *desc, uint32_t n_filters,
                                   uint64_t n_words, uint8_t *out_region, uint64_t region_cap, uint64_t *out_sec_off);
/* Device time of the most recent k_encode_payload + k_crc_sections pair (the slowest device's). */
BSG_API int32_t bsg_last_encode_ms(bsg_ctx *ctx, float *encode_ms);

/* ---- pinned host memory ----
 * Every entry point accepts ordinary (pageable) host pointers.  A host that marshals its rows, entries or sections
 * straight into a buffer from bsg_pinned_alloc (cgo: C memory, fill it through unsafe.Slice) gets asynchronous DMA at
 * the link's full rate instead of the driver's staged copy (measured: bsg_ingest_* of 1 M rows / 254 MB end to end
 * 23 -> 11 ms). */
BSG_API int32_t bsg_pinned_alloc(bsg_ctx *ctx, uint64_t n_bytes, void **out_ptr);
BSG_API int32_t bsg_pinned_free(bsg_ctx *ctx, void *ptr);
/* Page-lock memory the caller already owns (C memory, an mmap of a file or of shared memory; never Go-heap memory). */
BSG_API int32_t bsg_host_register(bsg_ctx *ctx, void *ptr, uint64_t n_bytes);
BSG_API int32_t bsg_host_unregister(bsg_ctx *ctx, void *ptr);

/* ---- device ingest: rows -> distinct bloom entries -> exact counts -> bitsets ----
 * Replaces, on the flush / merge worker, the reference's per-row host loop
 *   bloomEntrySets.indexRow (ingest.go:55-89: pathWalker.walk row_matcher.go:51-135, leafTokenInput
 *   tokenizer.go:120-133, BasicWhitespaceLowerTokenizer tokenizer.go:141-143, addFieldToken ingest.go:95-102),
 *   unionInto (ingest.go:105-115), counts (ingest.go:117-123) and buildFilters' AddString loop (ingest.go:127-145)
 * for the DEFAULT tokenizer (the reference's own fast-path check, row_matcher.go:37-40; custom tokenizers
 * keep the host path).  A "set" is one partition buffer's bloomEntrySets; set s owns rows
 * [set_first_row[s], set_first_row[s+1]).  A "parent" is a file-level union (flush.go:221,253);
 * parent_of_set[s] names it or is 0xFFFFFFFF.  Tables are indexed t = set * 3 + kind with parents
 * numbered after the sets (set index n_sets + p).
 *
 * Call order:  bsg_ingest_rows -> [bsg_ingest_fallback_rows -> host walker -> bsg_ingest_add_entries]
 *              -> bsg_ingest_finish (exact distinct counts; the caller sizes (m, k) with its own
 *              EstimateParameters) -> bsg_ingest_build -> bsg_ingest_free.
 * The device walker finishes rows of valid UTF-8 (see ingest.hip.h for the few exceptions); every other
 * row is reported by bsg_ingest_fallback_rows and MUST be walked by the host and added back with
 * bsg_ingest_add_entries before bsg_ingest_finish, or its entries are missing. */
typedef struct bsg_ingest_stats {
    uint32_t n_rows;
    uint32_t n_fallback_rows;  /* rows left to the host walker */
    uint32_t table_grows;      /* distinct-entry tables that had to be enlarged (x4 + rehash) */
    uint32_t reserved;
    uint64_t row_bytes;
    uint64_t table_bytes;      /* HBM held by the distinct-entry tables */
    float ms_walk;             /* k_ingest_rows dispatch time (sum over re-runs after a table grew) */
    float ms_union;            /* k_ingest_union into the parents */
    float ms_build;            /* k_build_sets (+ k_bin_* for bitsets beyond LDS): first dispatch start to last dispatch end */
    float ms_encode;           /* k_encode_payload + k_crc_sections (bsg_ingest_build_sections) */
} bsg_ingest_stats;

/* flags for bsg_ingest_rows.  BSG_INGEST_TRUSTED_JSON: every row is known to be valid JSON (it came out of
 * json.Marshal, or the caller validated it): the device skips its validation pass.  Without the flag a row the
 * device walker hands back has inserted nothing; with it, such a row may already have inserted some of its
 * entries — all of which the host walker inserts again for a valid row, so the result is the same. */
#define BSG_INGEST_TRUSTED_JSON 1u

/* rows: marshaled JSON, row r = rows[row_off[r] .. row_off[r+1]) without a trailing newline.
 * slots_hint: optional initial table capacities [n_sets * 3] (0 = default: 256 for fields, 4 per row for
 * tokens and field::tokens); tables grow on demand, a hint only saves the re-run. */
BSG_API int32_t bsg_ingest_rows(bsg_ctx *ctx, const uint8_t *rows, const uint64_t *row_off, uint32_t n_rows,
                                const uint32_t *set_first_row, uint32_t n_sets, const uint32_t *parent_of_set,
                                uint32_t n_parents, const uint32_t *slots_hint, uint32_t flags, uint64_t *out_ingest_id);
/* bsg_ingest_rows uploads the rows in chunks: the first of about this many bytes (default 64 MiB, 0 restores it), each
 * later one twice the one before up to four times this; the copy of chunk i+1 overlaps the walk of chunk i. */
BSG_API int32_t bsg_set_ingest_chunk(bsg_ctx *ctx, uint64_t bytes);
/* Row indices (ascending) the host walker must finish; rows_out may be NULL to query the count. */
BSG_API int32_t bsg_ingest_fallback_rows(bsg_ctx *ctx, uint64_t ingest_id, uint32_t *rows_out, uint32_t cap,
                                         uint32_t *n_out);
/* Entries produced by the host walker for the fallback rows: packed like bsg_hash_entries, each tagged
 * with its set (< n_sets) and kind. */
BSG_API int32_t bsg_ingest_add_entries(bsg_ctx *ctx, uint64_t ingest_id, const uint8_t *bytes, const uint32_t *offsets,
                                       uint32_t n_entries, const uint32_t *set_of_entry, const uint32_t *kind_of_entry);
/* Unions the sets into their parents and returns the exact distinct counts
 * out_counts[(n_sets + n_parents) * 3] (bloomEntrySets.counts).  out_status[n_sets + n_parents] (may be
 * NULL): 0 ok, 2 = the set cannot be represented and must be rebuilt on the host path: it met an entry whose
 * base hashes contain a zero word (probability 2^-62 per entry), a slot claim that was never completed (a wave
 * context-switched out for ~seconds between its CAS and its stores), or two DIFFERENT entries with the same four base hashes
 * (a MurmurHash3 state collision, constructible for entries of >= 24 bytes: every entry also carries a 64-bit fingerprint
 * under a per-context secret key, so the pair is noticed instead of being counted once — Go's map would count two). */
BSG_API int32_t bsg_ingest_finish(bsg_ctx *ctx, uint64_t ingest_id, uint64_t *out_counts, uint32_t *out_status);
/* desc[(n_sets + n_parents) * 3]: geometry and output offsets of every table's filter (m == 0 skips it);
 * out_words as bsg_build. */
BSG_API int32_t bsg_ingest_build(bsg_ctx *ctx, uint64_t ingest_id, const bsg_filter_desc *desc, uint64_t *out_words,
                                 uint64_t n_words);
/* As bsg_ingest_build, but the words never leave the device: every set's three filters are serialised there as one
 * filter section (see bsg_build_sections) and only the section bytes come back.  desc gives (m, k) per table (word_off
 * is ignored: the device lays the words out itself); out_sec_off[n_sets + n_parents + 1].
 * out_sets_arena_id / out_parents_arena_id (either may be NULL): the filters just built are also left RESIDENT as
 * probe arenas — "block" i of the first = set i, "block" p of the second = parent p — so the file that is being
 * written can be queried without its sections ever being uploaded or decoded again.  On a context opened on several
 * devices every device receives its blocks (b % n) device to device. */
BSG_API int32_t bsg_ingest_build_sections(bsg_ctx *ctx, uint64_t ingest_id, const bsg_filter_desc *desc, uint8_t *out_region,
                                          uint64_t region_cap, uint64_t *out_sec_off, uint64_t *out_sets_arena_id,
                                          uint64_t *out_parents_arena_id);
BSG_API int32_t bsg_ingest_stats_read(bsg_ctx *ctx, uint64_t ingest_id, bsg_ingest_stats *out);
BSG_API int32_t bsg_ingest_free(bsg_ctx *ctx, uint64_t ingest_id);

/* ---- final row test on the device (compileRowMatcher / matchRowBytes, row_matcher.go:257-626) ----
 * One expression over Field / Token / FieldToken conditions, evaluated for every row of the surviving blocks.
 * Condition i: cond_kinds[i] (BSG_KIND_*) and two strings packed like bsg_hash_entries' input — entry 2i = the FIELD
 * string (Field, FieldToken; empty for Token), entry 2i + 1 = the TOKEN string (Token, FieldToken; empty for Field):
 * cond_off[2 * n_conds + 1].  FieldToken is the (path, token) PAIR at one leaf, not the joined "path::token" key
 * (row_matcher.go:587).  prog_ops: the public postfix program over condition indices (BSG_OP_TERM i / AND n / OR n / TRUE /
 * FALSE; n_ops == 0 = nil expression = every row matches; a nil condition lowers to TRUE, an unknown condition or expression
 * type to FALSE, as evalMatcherNode does).
 * out_bits[ceil(n_rows / 64)]: bit r & 63 of word r >> 6 set <=> row r matches.  Rows the device cannot decide are listed
 * in out_fallback_rows (ascending; their bit is 0) and must be decided by the host matcher: rows outside the device
 * walker's envelope (see bsg_ingest_rows), and rows in which an emission has the base hashes of a condition string but
 * not its keyed fingerprint — a MurmurHash3 state collision, where only matchRowBytes' byte compare is exact.
 * Target tokens are never normalised (Token("ALICE") misses, row_matcher_test.go:99-100).
 * The rows of a large call are uploaded in chunks (bsg_set_ingest_chunk) on a copy stream while the chunk before is matched.
 * Limits: 64 conditions, expression depth 64 (BSG_E_UNSUPPORTED beyond). */
BSG_API int32_t bsg_match_rows(bsg_ctx *ctx, const uint8_t *rows, const uint64_t *row_off, uint32_t n_rows,
                               const uint8_t *cond_bytes, const uint32_t *cond_off, const uint32_t *cond_kinds, uint32_t n_conds,
                               const uint32_t *prog_ops, uint32_t n_ops,
                               uint64_t *out_bits, uint32_t *out_fallback_rows, uint32_t fallback_cap,
                               uint32_t *out_n_fallback);
/* Device time of the most recent k_match_rows dispatch (the slowest device's). */
BSG_API int32_t bsg_last_match_ms(bsg_ctx *ctx, float *match_ms);

#ifdef __cplusplus
}
#endif
#endif /* BLOOMGPU_H */
