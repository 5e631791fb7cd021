/* bloomgpu_lab.h — LAB switches of libbloomgpu.so: knobs for tools/, bench sweeps and tests.
 *
 * NOT part of the drop-in contract (include/bloomgpu.h): nothing here is needed to replace the reference's
 * seams, defaults are what the measurements under profiles/ chose, and a host binding (go/bloomgpu) must not include this
 * file — tools/check_go.py asserts that it does not.  Keys and meanings may change between rounds. */
#ifndef BLOOMGPU_LAB_H
#define BLOOMGPU_LAB_H

#include "bloomgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Lab knobs for tools/, bench sweeps and tests (key 1: compaction rounds of the many-term probe mode; key 2: HBM bytes a
 * binned build of a bitset beyond LDS may park its locations in, 0 = build it with global atomics; key 3: most distinct terms
 * of a synchronous batch of <= 256 queries that is answered by one dispatch, 0 = never; key 4: launches the decode of
 * bsg_arena_load_sections is split into, 1 = one launch after the whole copy; key 6: fewest locations (entries x k) from
 * which a bitset beyond LDS is built from binned locations instead of global atomics; keys 7 / 8: fewest entries / row bytes from
 * which a construct or match call on a context over several devices is cut into one part per device; key 9: 1 = file-level unions
 * through global hash tables instead of LDS partitions, key 10: start that partitioning 2^value x too coarse; key 11: evaluators per
 * tile of k_probe_eval — probe and program evaluation of few-term batches in ONE dispatch —, 0 = two dispatches, the default;
 * keys 12-17, 20-22, 24: the combiner of concurrent bsg_query calls — 12: 0 = every call alone, 1 = combine (default), 2 = lab, the caller's
 * preparation only, nothing probed; 13: cycles in flight (2; one more while cycles average > 32 calls); 15: (microseconds << 16) |
 * calls a collector waits for company (tests); 16: 3-term queries asked of one arena in a cycle from which it is streamed once for all of
 * them (8, for 35 KB of filters per block: scaled by the arena's bytes per block and the calls' distinct terms); 17: microseconds a queued caller polls while the context is quiet (60); 20: account the callers' processor time
 * (bsg_lab_query_cpu); 21: 0 = a cycle's job table is always uploaded (default 1: a table of <= ~4 KB rides in the kernel arguments); 22: workgroups of a lone call's dispatch beyond which its doorbell is a dispatch behind it (32); 24: bytes of survivor rows beyond which a cycle is served in parts (64 MB); 25: microseconds a collector polls its dispatch's doorbell before it sleeps on an event behind the dispatch (0 = adaptive, the default: 4 x the running mean of the polled waits within [50 us, 1 ms]); key 19: percent of a single-group device-resident run whose evaluation moves to a second stream (0 = off))) */
BSG_API int32_t bsg_set_lab(bsg_ctx *ctx, uint32_t key, uint64_t value);
/* Synchronous probes poll their stream for up to this long before they block (default 0: block at once).  A single
 * query's kernels finish in ~10 us; being woken from a blocking wait costs more than that. */
BSG_API int32_t bsg_set_spin_wait(bsg_ctx *ctx, uint32_t microseconds);
/* Groups of up to this many arenas ride fused (k_probe_fused: the probe of group i and the program evaluation of group
 * i-1 in one dispatch; default 4, 0 = never).  Larger groups run as k_probe_terms + k_eval_programs. */
BSG_API int32_t bsg_set_fuse_limit(bsg_ctx *ctx, uint32_t max_arenas);
/* Gather regime: a filter is read by <= terms * k sector gathers instead of being streamed into LDS when
 * terms * k * bytes_per_probe < its size (default 256; 0 = always stream). */
BSG_API int32_t bsg_set_gather_cost(bsg_ctx *ctx, uint32_t bytes_per_probe);

/* Lab (bsg_set_lab key 20 = 1 turns the accounting on): the CALLERS' own processor time, summed — out[0] profiled calls, out[1] ns
 * inside bsg_query's combiner path, of which out[2] up to the end of the wait (push, polling, the futex), out[3] waking other
 * callers, out[4] collecting (a collector's whole cycle).  tools/conc_lab.py prints them per call. */
BSG_API int32_t bsg_lab_query_cpu(bsg_ctx *ctx, uint64_t *out, int32_t reset);

/* With BSG_PROBE_TIMED, only every stride-th dispatch group is timestamped (default 1 = all). */
BSG_API int32_t bsg_set_timed_stride(bsg_ctx *ctx, uint32_t stride);

#ifdef __cplusplus
}
#endif
#endif /* BLOOMGPU_LAB_H */
