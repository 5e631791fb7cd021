"""The legs of bench.py beyond the headline (each a function returning the object that lands in bench_legs.json)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from . import common as C
from .common import (HBM_PEAK_GBPS, CLOCK_NOTE, C5_TOTAL_FILTERS, COLL_DEVICE, Prober, Ring, SharedHost, generate_blocks, pool_map,
                     _gen_rows, kernel_stats, dominant_kernel)


def build_leg(ctx, plan, B, rows, log, calls=6):
    """C3 (BASELINE configs[2]) through the product build path: bsg_build of the shard's pre-extracted entry sets -> 3 B bitsets.
    kernel_ms = MEDIAN of the warm calls' k_build dispatches (each timed by its own start/stop events); the first call of the
    process — which also pays the code object's load and the scratch's hipMalloc — is reported separately (first_call_ms).
    Algorithmic bytes: entry bytes + offsets + filter ranges + every bitset written once."""
    t0 = time.time()
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    ms = [ctx.last_kernel_ms()[0]]
    for _ in range(max(0, calls - 1)):
        again = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        ms.append(ctx.last_kernel_ms()[0])
        if not np.array_equal(again, words):
            sys.exit("two bsg_build calls over the same entries returned different bitsets")
    warm = ms[1:] or ms
    kernel_ms = float(np.median(warm))
    nbytes = len(plan.blob) + 4 * len(plan.off) + 4 * len(plan.fstart) + int(sum((int(m) + 63) // 64 * 8 for m in plan.desc["m"]))
    res = {"workload": "C3 flush-side build: %d blocks x %d rows -> %d filters, %d distinct entries" % (B, rows, 3 * B, len(plan.off) - 1),
           "kernel": "k_build", "kernel_ms": kernel_ms, "first_call_ms": ms[0], "warm_calls_ms": warm, "calls": len(ms),
           "algorithmic_bytes": nbytes, "achieved": nbytes / max(kernel_ms, 1e-6) / 1e6, "unit": "GB/s",
           "frac": nbytes / max(kernel_ms, 1e-6) / 1e6 / HBM_PEAK_GBPS, "entries_per_s": (len(plan.off) - 1) / max(kernel_ms, 1e-6) * 1e3}
    log("built %d filters (%.1f MB of bitsets) %d x in %.2fs; k_build warm median %.1f us = %.0f GB/s algorithmic (%.3f of HBM peak); first call %.1f us"
        % (3 * B, plan.n_words * 8 / 1e6, len(ms), time.time() - t0, kernel_ms * 1e3, res["achieved"], res["frac"], ms[0] * 1e3))
    return words, res


def ingest_leg(ctx, n_blocks, rows, seed, workers, plan, words, fpr, log, trusted=1):
    """C3 from the front of the path: the JSON rows of the first n_blocks blocks -> k_ingest_rows (walk, tokenize,
    hash, dedup) -> k_ingest_union (file-level sets) -> exact counts -> k_build_sets.  The bitsets must equal the ones
    bsg_build produced from the pre-extracted entry sets of the same blocks, bit for bit."""
    from bloomsearch_amd import ingest as I
    t0 = time.time()
    parts = pool_map(_gen_rows, [(b, rows, seed) for b in range(n_blocks)], workers)
    blob = np.frombuffer(b"".join(p[0] for p in parts), dtype=np.uint8)
    lens = np.concatenate([p[1] for p in parts])
    del parts
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    first = np.arange(n_blocks + 1, dtype=np.uint32) * rows
    t_gen = time.time() - t0
    t0 = time.time()
    ing = ctx.ingest_rows((blob, off), first, np.zeros(n_blocks, dtype=np.uint32), 1, flags=trusted)
    fb = ctx.ingest_fallback_rows(ing)
    counts, status = ctx.ingest_finish(ing, n_blocks + 1)
    desc, n_words = I.plan_desc(counts, fpr)
    got = ctx.ingest_build(ing, desc, n_words)
    t_e2e = time.time() - t0
    st = ctx.ingest_stats(ing)
    ctx.ingest_free(ing)
    # the same again with the rows marshalled into pinned host memory (bsg_pinned_alloc): the copy becomes a plain DMA
    pinned = ctx.pinned_array(len(blob))
    pinned[:] = blob
    pinned_out = ctx.pinned_array(n_words * 8)                 # the bitsets come back into page-locked memory too
    pinned_out[:] = 0
    t0 = time.time()
    ing2 = ctx.ingest_rows((pinned, off), first, np.zeros(n_blocks, dtype=np.uint32), 1, flags=trusted)
    counts2, _ = ctx.ingest_finish(ing2, n_blocks + 1)
    desc2, n_words2 = I.plan_desc(counts2, fpr)
    got2 = ctx.ingest_build(ing2, desc2, n_words2, out=pinned_out.view(np.uint64))[:n_words2]
    t_e2e_pinned = time.time() - t0
    st_first, st = st, ctx.ingest_stats(ing2)     # kernel times of record: the second run (the first one's may hold a hipMalloc of the
    ctx.ingest_free(ing2)                         # buffer pool inside a kernel's timestamps: k_build_sets 28 instead of 3.2 ms, seen once in ten runs)
    got2 = got2.copy()
    ctx.pinned_free(pinned_out)
    if not (np.array_equal(counts2, counts) and np.array_equal(got2, got)):
        sys.exit("device ingest from pinned rows differs from the pageable run")
    if len(fb) or status.any():
        sys.exit("device ingest handed back %d synthetic rows / flagged a set" % len(fb))
    for i in range(n_blocks * 3):
        d, e = desc[i], plan.desc[i]
        nw = (int(d["m"]) + 63) // 64
        if (int(d["m"]), int(d["k"])) != (int(e["m"]), int(e["k"])) or not np.array_equal(
                got[int(d["word_off"]): int(d["word_off"]) + nw], words[int(e["word_off"]): int(e["word_off"]) + nw]):
            sys.exit("device ingest filter %d differs from bsg_build of the same block's entry sets" % i)
    kern_ms = st.ms_walk + st.ms_union + st.ms_build
    n_rows = n_blocks * rows
    log("device ingest: %d rows (%.0f MB JSON) walk %.2f ms + union %.2f ms + build %.2f ms = %.1f M rows/s on-device; "
        "%.3fs end to end incl. the chunked H2D overlapping the walk (%.3fs = %.2f x the kernels with rows and bitsets in page-locked memory; row generation %.1fs); "
        "filters bit-identical to bsg_build"
        % (n_rows, st.row_bytes / 1e6, st.ms_walk, st.ms_union, st.ms_build, n_rows / kern_ms / 1e3, t_e2e, t_e2e_pinned,
           t_e2e_pinned * 1e3 / kern_ms, t_gen))
    # the final row test (BASELINE configs[0]'s query, FieldToken("level", "error"), row_matcher.go) over the same rows on
    # the device; truth = the generator's own draws
    from bloomsearch_amd import query as Q, synth
    t0 = time.time()
    hits, mfb = ctx.match_rows((blob, off), Q.CompiledMatcher(Q.FieldToken("level", "error")))
    t_match = time.time() - t0
    match_ms = ctx.last_match_ms()
    t0 = time.time()
    hits_p, _ = ctx.match_rows((pinned, off), Q.CompiledMatcher(Q.FieldToken("level", "error")))      # the same rows in page-locked memory
    t_match_pinned = time.time() - t0
    ctx.pinned_free(pinned)
    if not np.array_equal(hits_p, hits):
        sys.exit("device row matcher: pinned and pageable rows disagree")
    truth = np.concatenate([synth.draws(b * rows, rows, seed)["level"] == synth.LEVELS.index("error") for b in range(n_blocks)])
    if len(mfb) or not np.array_equal(hits, truth):
        sys.exit("device row matcher disagrees with the generator's ground truth")
    log("device row match: %d rows in %.2f ms = %.0f M rows/s (%.0f GB/s of JSON), %d matches; %.3fs end to end incl. the chunked H2D under the kernel "
        "(%.3fs with the rows in page-locked memory)"
        % (n_rows, match_ms, n_rows / match_ms / 1e3, st.row_bytes / match_ms / 1e6, int(hits.sum()), t_match, t_match_pinned))
    match = {"workload": "final row test FieldToken(level, error) over the same %d rows" % n_rows, "kernel": "k_match_rows",
             "kernel_ms": match_ms, "rows_per_s_device": n_rows / match_ms * 1e3, "row_gb_per_s": st.row_bytes / match_ms / 1e6,
             "matches": int(hits.sum()), "end_to_end_s_incl_h2d": t_match, "end_to_end_s_incl_h2d_pinned_rows": t_match_pinned,
             "upload": "rows travel in chunks of 64, 128, then 256 MiB on a copy stream while the chunk before is being matched",
             "check": "equals the generator's draws row for row"}
    return {"workload": "C3 from rows: %d blocks x %d JSON rows -> %d block filters + 3 file-level filters" % (n_blocks, rows, 3 * n_blocks),
            "match": match,
            "kernels": {"k_ingest_rows_ms": st.ms_walk, "k_union_partitions_ms": st.ms_union, "k_build_sets_ms": st.ms_build},
            "kernels_first_run": {"k_ingest_rows_ms": st_first.ms_walk, "k_union_partitions_ms": st_first.ms_union, "k_build_sets_ms": st_first.ms_build},
            "rows": n_rows, "row_bytes": int(st.row_bytes), "rows_per_s_device": n_rows / kern_ms * 1e3,
            "row_gb_per_s_walk": st.row_bytes / max(st.ms_walk, 1e-6) / 1e6, "end_to_end_s_incl_h2d": t_e2e,
            "end_to_end_s_incl_h2d_pinned_rows": t_e2e_pinned, "end_to_end_over_kernels_pinned": t_e2e_pinned * 1e3 / kern_ms,
            "upload": "rows travel in chunks of 64, 128, then 256 MiB on a copy stream while the chunk before is being walked",
            "end_to_end": "bsg_ingest_rows + bsg_ingest_finish + EstimateParameters on the host + bsg_ingest_build with the bitsets copied back; "
                          "the pinned figure has rows and bitsets in memory from bsg_pinned_alloc",
            "table_bytes": int(st.table_bytes), "table_grows": int(st.table_grows), "fallback_rows": int(len(fb)),
            "distinct_entries": int(counts[:n_blocks].sum()), "file_level_distinct": [int(x) for x in counts[n_blocks]],
            "check": "bitsets and (m, k) identical to bsg_build of the same blocks' entry sets"}


def multi_device_context_leg(ctx, device_ids, plan, words, block_ids, rows, seed, fpr, terms, ops, poff, got, log, res=None):
    """The OTHER way the library spans GPUs: ONE process, one context over several devices (what the Go engine opens:
    GPUDevices = [0 .. N-1]).  bench.py's contract is one process per GPU, so this in-process path is otherwise only ever run
    on one physical GPU (contexts that name device 0 several times).  Rank 0 runs it once, after every timed leg, when the
    job has several GPUs in sight: arena sharded block b -> device b % N, the C2 batch probed there (survivors interleaved on
    the host), one interactive query, the C3 build cut into one part per device (parts on threads, each over its own PCIe
    link), a device ingest whose parents are merged across devices (peer copies) and whose sections become resident arenas,
    and the fixed-geometry OR with the partials moved device to device.  Everything is compared with the single-device
    context's results, which the legs above compared with the oracle.  Never fatal: the outcome goes into the line."""
    from bloomsearch_amd import ingest as I, query as Q
    from bloomsearch_amd.gpu import Context
    B = len(block_ids)
    res = {} if res is None else res           # filled stage by stage: if a stage never returns, the watchdog's line still holds the others
    res.update({"devices": [int(d) for d in device_ids], "stages_done": [], "stage_running": "open"})

    def stage(name):
        if res["stage_running"] not in ("open",):
            res["stages_done"].append(res["stage_running"])
        res["stage_running"] = name
    with Context(tuple(device_ids)) as m:
        res["peer_access"] = m.peer_access().tolist()        # 1: direct xGMI peer access; 0: copies between the pair are staged through the host
        stage("probe")
        aid = m.arena_load(words, plan.desc)
        t0 = time.perf_counter()
        if not np.array_equal(m.probe(aid, B, terms, ops, poff), got):
            raise RuntimeError("survivors of the multi-device context differ from the single-device context's")
        res["probe_wall_ms"] = (time.perf_counter() - t0) * 1e3
        # the host-side gather as survivor ROWS: every device writes its shards' rows into its slice of one page-locked buffer, the
        # host merges a (file, query)'s rows into global block ids (bsg_survivor_rows_list) — against the dense bitsets + interleave
        stage("rows")
        bid_m = m.batch_create(terms, ops, poff)
        NQ = len(poff) - 1
        rw, hw = m.survivor_rows_size([aid], bid_m)
        r_buf = m.pinned_array(max(rw, 1) * 8).view(np.uint64)
        h_buf = m.pinned_array(hw * 4).view(np.uint32)
        dense_buf = m.pinned_array(NQ * ((B + 63) // 64) * 8).view(np.uint64)
        t_rows, t_dense = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            m.probe_many_rows([aid], bid_m, r_buf, h_buf)
            t_rows.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            m.probe_many_into([aid], bid_m, dense_buf)
            t_dense.append(time.perf_counter() - t0)
        sel = np.sort(np.random.default_rng(99).choice(NQ, size=min(NQ, 64), replace=False))
        for q in sel:
            ids = m.survivor_rows_list([aid], bid_m, r_buf, h_buf, 0, int(q), B).astype(np.int64)
            bits = np.zeros_like(got[q])
            np.bitwise_or.at(bits, ids >> 6, np.uint64(1) << (ids & 63).astype(np.uint64))
            if not np.array_equal(bits, got[q]):
                raise RuntimeError("survivor rows of the multi-device context do not merge to the single-device context's survivors (query %d)" % q)
        if not np.array_equal(dense_buf.reshape(NQ, -1), got):
            raise RuntimeError("dense survivors of the multi-device context differ")
        tags = np.bincount(h_buf >> 30, minlength=4)
        cnt = h_buf & np.uint32(0x3FFFFFFF)
        nd_m = len(device_ids)
        res["rows"] = {"api": "bsg_probe_many_rows on the %d-device context + bsg_survivor_rows_list" % nd_m,
                       "wall_ms_rows": float(np.median(t_rows[1:])) * 1e3, "wall_ms_dense_bitsets_interleaved_on_host": float(np.median(t_dense[1:])) * 1e3,
                       "bytes_written_by_the_devices": int(4 * hw + 4 * cnt[(h_buf >> 30) == 2].sum() + 8 * ((B // nd_m + 63) // 64) * int(((h_buf >> 30) == 3).sum())),
                       "dense_bytes": int(NQ * ((B + 63) // 64) * 8), "rows_by_tag_none_all_list_dense": [int(x) for x in tags],
                       "check": "%d randomly chosen queries: the %d shards' rows merge to the single-device survivors" % (len(sel), nd_m)}
        m.pinned_free(r_buf.view(np.uint8)); m.pinned_free(h_buf.view(np.uint8)); m.pinned_free(dense_buf.view(np.uint8))
        m.batch_free(bid_m)
        stage("bsg_query")
        one = Q.compile_queries([Q.And(Q.FieldToken("level", "error"), Q.FieldToken("service", "payment"), Q.FieldToken("nested.region", "region-3"))])
        a1 = ctx.arena_load(words, plan.desc)
        same = np.array_equal(m.query([aid], [B], one)[0], ctx.query([a1], [B], one)[0])
        ctx.arena_free(a1)
        m.arena_free(aid)
        if not same:
            raise RuntimeError("bsg_query on the multi-device context differs")
        # the C3 build, one part per device
        stage("build")
        t0 = time.perf_counter()
        w_m = m.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        t_m = time.perf_counter() - t0
        t0 = time.perf_counter()
        w_s = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        t_s = time.perf_counter() - t0
        if not (np.array_equal(w_m, words) and np.array_equal(w_s, words)):
            raise RuntimeError("bitsets of the sharded build differ")
        res["build_wall_ms"] = {"one_device": t_s * 1e3, "sharded": t_m * 1e3,
                                "note": "bsg_build of %d entries incl. the upload of the entry bytes and the bitsets' way back" % (len(plan.off) - 1)}
        # device ingest of the first blocks' rows: parts per device, parents merged across devices, sections + resident arenas
        stage("ingest")
        nb = min(B, 64)
        parts = [_gen_rows((int(block_ids[b]), rows, seed)) for b in range(nb)]
        blob = np.frombuffer(b"".join(p[0] for p in parts), dtype=np.uint8)
        off = np.zeros(sum(len(p[1]) for p in parts) + 1, dtype=np.uint64)
        np.cumsum(np.concatenate([p[1] for p in parts]), out=off[1:])
        first = np.arange(nb + 1, dtype=np.uint32) * rows
        outs = []
        for c in (m, ctx):
            ing = c.ingest_rows((blob, off), first, np.zeros(nb, dtype=np.uint32), 1, flags=1)
            counts, status = c.ingest_finish(ing, nb + 1)
            desc, n_words = I.plan_desc(counts, fpr)
            secs, sets_arena, parents_arena = c.ingest_build_sections(ing, desc, arenas=True)
            c.ingest_free(ing)
            surv = c.probe(sets_arena, nb, terms, ops, poff)
            c.arena_free(sets_arena)
            c.arena_free(parents_arena)
            outs.append((counts, status, [bytes(x) for x in secs], surv))
        if not (np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
                and np.array_equal(outs[0][3], outs[1][3])):
            raise RuntimeError("device ingest on the multi-device context differs (counts, section bytes or the resident arena's survivors)")
        if nb == 64 and not np.array_equal(outs[0][3][:, 0], got[:, 0]):      # blocks 0..63 are the first survivor word of the loaded arena
            raise RuntimeError("the resident arena of the first blocks answers differently from the loaded arena")
        # fixed-geometry OR, partials device to device
        stage("or_reduce")
        rng = np.random.default_rng(5)
        mm, nblk = 1000003, 96
        nw = (mm + 63) // 64
        stride = (nw + 15) // 16 * 16
        from bloomsearch_amd import _lib
        d2 = np.zeros(nblk * 3, dtype=_lib.DESC_DTYPE)
        w2 = np.zeros(nblk * stride, dtype=np.uint64)
        for b in range(nblk):
            d2[b * 3 + 1] = (b * stride, mm, 7, 0)
            w2[b * stride: b * stride + nw] = rng.integers(0, 1 << 63, nw, dtype=np.uint64) & rng.integers(0, 1 << 63, nw, dtype=np.uint64)
            w2[b * stride + nw - 1] &= np.uint64((1 << (mm & 63)) - 1)
        a2 = m.arena_load(w2, d2)
        got_or = m.or_reduce(a2, 1, nw)
        m.arena_free(a2)
        if not np.array_equal(got_or, np.bitwise_or.reduce(w2.reshape(nblk, stride)[:, :nw], axis=0)):
            raise RuntimeError("bsg_or_reduce across the context's devices differs from numpy's OR")
        res["device_calls"] = [int(x) for x in m.device_calls()]
        stage("close")
    res["stages_done"].append("close")
    res["stage_running"] = None
    res["check"] = ("probe (batch + one bsg_query), bsg_build, device ingest -> sections + resident arenas, bsg_or_reduce: identical to the "
                    "single-device context on %d devices in one process" % len(device_ids))
    log("multi-device context over devices %s: ok (build %.0f ms sharded vs %.0f ms on one device)" % (list(device_ids), t_m * 1e3, t_s * 1e3))
    return res


def or_reduce_leg(ctx, plan, B, fpr, n_union, world, log):
    """BASELINE configs[4] / SURVEY C5: OR-reduce of this rank's B fixed-geometry token filters into one partial
    file-level bitset (k_or_reduce_blocks), then — for world > 1 — the one real exchange of the path: all_gather of the
    partials over RCCL + a local OR (k_or_words).  Geometry = EstimateParameters(n_union, fpr) for every block, which
    is what makes OR_b build(S_b, m, k) == build(U S_b, m, k) hold."""
    import torch
    from bloomsearch_amd import parallel as P
    from bloomsearch_amd._lib import DESC_DTYPE
    from bloomsearch_amd.gpu import estimate_parameters
    m, k = estimate_parameters(n_union, fpr)
    nw = (m + 63) // 64
    stride = (nw + 15) // 16 * 16
    desc = np.zeros(B * 3, dtype=DESC_DTYPE)
    for b in range(B):
        desc[b * 3 + 1] = (b * stride, m, k, 0)        # token filters only; field / field::token left nil
    t0 = time.time()
    words = ctx.build(plan.blob, plan.off, plan.fstart, desc, B * stride)
    # BASELINE configs[4] reduces 10 000 block filters in all: at N > 1 this rank holds its 10 000 / N of them (1 250 at N = 8) —
    # the B built ones, then address-distinct copies of them in order (a copy adds no bit to the OR; every byte is still read from
    # an address of its own).  N = 1 keeps the B filters of its own blocks.
    n_f = B
    if world > 1:
        import torch.distributed as dist
        n_f = C5_TOTAL_FILTERS // world + (1 if dist.get_rank() < C5_TOTAL_FILTERS % world else 0)
        reps = (n_f + B - 1) // B
        words = np.concatenate([words[: B * stride]] * reps)[: n_f * stride]
        desc = np.zeros(n_f * 3, dtype=DESC_DTYPE)
        for b in range(n_f):
            desc[b * 3 + 1] = (b * stride, m, k, 0)
    aid = ctx.arena_load(words, desc)
    t_setup = time.time() - t0
    out = torch.zeros(nw, dtype=torch.int64, device="cuda")
    ms = []
    for _ in range(12):
        ctx.or_reduce_dev(aid, 1, out.data_ptr(), nw)
        ms.append(ctx.last_or_ms())
    local_ms = float(np.median(ms[2:]))
    # check: the OR equals one build of the union's entries at the same geometry (different kernel, same arithmetic),
    # and holds every token of the first and last block (oracle bit tests are in tests/test_gpu_parity.py)
    off64 = plan.off.astype(np.int64)
    blobs, lens = [], []
    for blk in range(B):                                  # a block's token entries are contiguous in the plan's blob
        e0, e1 = int(plan.fstart[blk * 3 + 1]), int(plan.fstart[blk * 3 + 2])
        blobs.append(plan.blob[off64[e0]: off64[e1]])
        lens.append(np.diff(off64[e0: e1 + 1]))
    ublob = np.concatenate(blobs)
    lens = np.concatenate(lens)
    idx = lens                                            # (one entry per element)
    uoff = np.zeros(len(lens) + 1, dtype=np.uint32)
    np.cumsum(lens, out=uoff[1:])
    udesc = np.zeros(1, dtype=DESC_DTYPE)
    udesc[0] = (0, m, k, 0)
    want = ctx.build(ublob, uoff, np.asarray([0, len(idx)], dtype=np.uint32), udesc, stride)[:nw]
    got = out.cpu().numpy().view(np.uint64)
    if not np.array_equal(got, want):
        sys.exit("OR-reduce of the block filters differs from the build of the union at the same geometry")
    res = {"workload": "C5 OR-reduce: %d fixed-geometry token filters on this GPU%s (m = %d bits, k = %d; %d distinct entries) -> one partial bitset"
                       % (n_f, " of %d over %d ranks" % (C5_TOTAL_FILTERS, world) if world > 1 else "", m, k, len(idx)),
           "filters_this_rank": n_f, "filters_total": C5_TOTAL_FILTERS if world > 1 else n_f,
           "kernel": "k_or_reduce_blocks", "kernel_ms": local_ms, "algorithmic_bytes": n_f * nw * 8 + nw * 8,
           "achieved": (n_f * nw * 8 + nw * 8) / max(local_ms, 1e-6) / 1e6, "unit": "GB/s", "bound": "hbm",
           "check": "equals bsg_build(union of the blocks' entries, m, k) bit for bit"}
    res["frac"] = res["achieved"] / HBM_PEAK_GBPS
    state = {"aid": aid, "out": out, "got": got, "nw": nw} if world > 1 and COLL_DEVICE() == "cuda" else None
    if state is None:
        ctx.arena_free(aid)
    log("OR-reduce: %d filters x %.0f KB in %.1f us = %.0f GB/s (%.0f%% of peak); setup %.1fs"
        % (n_f, nw * 8 / 1e3, local_ms * 1e3, res["achieved"], 100 * res["frac"], t_setup))
    return res, state


def or_exchange_leg(ctx, res, state, world, log):
    """The exchange half of C5, inside the library (bsg_or_allreduce: slice-wise ncclSend/ncclRecv + k_or_words + ncclAllGather over xGMI); the unique id
    travels over the harness' own channel.  Runs LAST and under a watchdog (main): a collective that never returns must not
    take the probe measurement down with it."""
    import torch
    aid, out, got, nw = state["aid"], state["out"], state["got"], state["nw"]
    try:
        import torch.distributed as dist
        from bloomsearch_amd.gpu import Context
        box = [Context.comm_unique_id() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(box[0], dist.get_rank(), world)
        n_seen, rank_seen, from_lib = ctx.comm_info()          # ncclCommCount / ncclCommUserRank of the library's own communicator
        res["n_ranks_seen_by_rccl"] = n_seen
        res["n_ranks_source"] = "ncclCommCount" if from_lib else "bsg_comm_init arguments (the bound library lacks ncclCommCount)"
        if n_seen != world or rank_seen != dist.get_rank():
            raise RuntimeError("RCCL sees rank %d of %d, the job is rank %d of %d" % (rank_seen, n_seen, dist.get_rank(), world))
        ts = []
        for _ in range(6):
            part = out.clone()
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.or_allreduce_dev([part.data_ptr()], nw)
            ts.append(time.perf_counter() - t1)
        full = ctx.or_allreduce(aid, 1, nw)            # the whole operation: local OR + exchange + copy out
        # a wrong result is reported in the line (allreduce_error) like a failed call: this leg is the extension of the path, and
        # it must not take the probe measurement down with it
        if not np.array_equal(full, part.cpu().numpy().view(np.uint64)):
            raise RuntimeError("bsg_or_allreduce and bsg_or_allreduce_dev disagree")
        # every rank must hold every rank's bits: the local partial is a subset of the result, and all ranks hold the same words
        if np.any(got & ~full):
            raise RuntimeError("OR all-reduce lost bits of this rank's partial")
        if world > 1:
            digest = torch.tensor([int(np.bitwise_xor.reduce(full) >> np.uint64(1)), int(full.sum(dtype=np.uint64) >> np.uint64(1))],
                                  dtype=torch.int64, device=COLL_DEVICE())
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                raise RuntimeError("the ranks hold different results after the OR all-reduce")
            res["allreduce_check"] = "every rank's partial is a subset of the result; xor / sum digests of the result equal on all %d ranks" % world
        ctx.comm_destroy()
        res["allreduce_ms"] = float(np.median(ts[1:])) * 1e3
        from bloomsearch_amd import parallel as P
        wire = P.or_allreduce_wire_bytes(nw, world)      # 2 (world - 1) / world x S: slice exchange + all-gather of reduced slices
        res["allreduce_wire_bytes_in_per_gpu"] = wire
        res["allreduce_wire_bytes_allgather_of_partials"] = (world - 1) * nw * 8      # what round 2's schedule moved
        res["allreduce_gbps_in_per_gpu"] = wire / max(res["allreduce_ms"], 1e-9) / 1e6
        res["allreduce"] = ("bsg_or_allreduce_dev: grouped ncclSend/ncclRecv of %d slices (slice j -> rank j) + k_or_words + ncclAllGather of the "
                            "reduced slices (RCCL over xGMI), inside libbloomgpu" % world)
        log("OR all-reduce over %d ranks: %.2f ms" % (world, res["allreduce_ms"]))
    except Exception as exc:  # noqa: BLE001 - reported, not swallowed
        res["allreduce_error"] = repr(exc)
        log("OR all-reduce failed: %r" % (exc,))
    ctx.arena_free(aid)


def q1_latency(ctx, arena, B, n_terms_hash, log):
    """SURVEY 8d C2's Q = 1 case: one 3-term And(FieldToken) query against the 1 000-block arena, survivors returned
    to the host — the latency a single interactive query sees — in both regimes: bitsets streamed into LDS
    (35 MB for 30 bit tests per block) and gathered (<= terms x k sector reads per block)."""
    from bloomsearch_amd import _lib, query as Q, synth
    cb = Q.compile_queries([Q.And(Q.FieldToken("level", "error"), Q.FieldToken("service", "payment"),
                                  Q.FieldToken("nested.region", "region-3"))])
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=_lib.TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    bid = ctx.batch_create(terms, ops, poff)
    res = {}
    first = None
    out = np.zeros((1, (B + 63) // 64), dtype=np.uint64)
    ids = [arena]

    def lat_loop(n):
        lat = []
        for _ in range(n):
            t0 = time.perf_counter()
            ctx.probe_many_into(ids, bid, out.reshape(-1))
            lat.append(time.perf_counter() - t0)
        return lat
    # one_dispatch: k_probe_direct (bit tests + program + survivors into page-locked memory in one launch); the other three
    # are the two-kernel path it replaces for such batches (lab key 3 = 0), gathered or streamed bitsets
    for name, direct, cost, spin in (("one_dispatch", 16, 256, 0), ("one_dispatch_spin_wait", 16, 256, 100), ("gather", 0, 256, 0),
                                     ("gather_spin_wait", 0, 256, 100), ("stream", 0, 0, 0)):
        ctx.set_lab(3, direct)
        ctx.set_gather_cost(cost)
        ctx.set_spin_wait(spin)
        lat_loop(50)
        lat = lat_loop(300)                                 # wall latency of a synchronous query, no timestamps
        ctx.timing_read(reset=True)
        for _ in range(32):
            got = ctx.probe_batch(arena, bid, 1, B, flags=_lib.PROBE_TIMED)
        tm = ctx.timing_read()
        if first is None:
            first = got
        if not (np.array_equal(first, got) and np.array_equal(first, out)):
            sys.exit("Q=1: the one-dispatch, gathered and streamed probes disagree")
        res[name] = {"latency_us_median": float(np.median(lat)) * 1e6, "latency_us_p90": float(np.percentile(lat, 90)) * 1e6}
        if direct:
            res[name]["k_probe_direct_us"] = tm.ms_fused_kernel / max(tm.n_fused, 1) * 1e3
        elif tm.n_folded:
            res[name]["k_probe_eval_us"] = tm.ms_folded_kernel / tm.n_folded * 1e3       # one dispatch: bit tests + programs per tile
        else:
            res[name]["k_probe_terms_us"] = tm.ms_terms_kernel / max(tm.n_probes, 1) * 1e3
            res[name]["k_eval_programs_us"] = tm.ms_eval_kernel / max(tm.n_eval, 1) * 1e3
    ctx.set_lab(3, 16)
    ctx.set_spin_wait(0)
    ctx.set_gather_cost(256)
    ctx.batch_free(bid)
    # What a synchronous Query() really pays, STRINGS in -> survivors out, nothing prepared beforehand:
    #   bsg_query       one call: terms hashed on the host, hashes + program in the kernel arguments, one dispatch, doorbell
    #   three_calls     the round-2 overlay path: bsg_hash_entries (a launch + sync for 3 strings) + bsg_batch_create (uploads) +
    #                   bsg_probe_many (+ bsg_batch_free)
    def e2e_query(n):
        lat = []
        for _ in range(n):
            t0 = time.perf_counter()
            ctx.query(ids, [B], cb, out.reshape(-1))
            lat.append(time.perf_counter() - t0)
        return lat

    def e2e_three_calls(n):
        lat = []
        for _ in range(n):
            t0 = time.perf_counter()
            t3 = np.zeros(len(cb.term_strings), dtype=_lib.TERM_DTYPE)
            t3["h"] = ctx.hash_strings(cb.term_strings)
            t3["kind"] = kinds
            b3 = ctx.batch_create(t3, ops, poff)
            ctx.probe_many_into(ids, b3, out.reshape(-1))
            ctx.batch_free(b3)
            lat.append(time.perf_counter() - t0)
        return lat
    for name, fn in (("bsg_query", e2e_query), ("three_calls", e2e_three_calls)):
        fn(50)
        lat = fn(300)
        if not np.array_equal(first, out):
            sys.exit("Q=1: %s disagrees with the probe of the prepared batch" % name)
        res["end_to_end_" + name] = {"latency_us_median": float(np.median(lat)) * 1e6, "latency_us_p90": float(np.percentile(lat, 90)) * 1e6}
    res["end_to_end_bsg_query"]["note"] = ("strings in -> survivors out in ONE call (bsg_query): 3 terms hashed on the host, hashes + lowered program in the "
                                          "kernel arguments of one k_query_direct dispatch, survivors in page-locked memory, doorbell; measured from Python "
                                          "(ctypes adds ~2 us per call)")
    res["end_to_end_three_calls"]["note"] = "bsg_hash_entries + bsg_batch_create + bsg_probe_many + bsg_batch_free per query (what one_dispatch's figure leaves out)"
    k = 10
    alg = k * 8 * len(terms) * B                      # SURVEY 8d gather regime: k x 8 B per (block, term) probe
    g = res["gather"]
    g["algorithmic_bytes"] = alg
    g_us = g.get("k_probe_eval_us") or g.get("k_probe_terms_us")
    g["achieved"] = alg / (g_us * 1e-6) / 1e9
    g["frac_of_hbm_peak"] = g["achieved"] / HBM_PEAK_GBPS
    g["note"] = "8-byte words out of 64-byte sectors: 12.5% of peak is the ceiling of this regime; at Q = 1 the kernel is launch-latency-bound"
    o = res["one_dispatch"]
    o["algorithmic_bytes"] = alg
    o["note"] = "k_probe_direct: one launch tests the bits, runs the program and writes the survivors into page-locked host memory"
    log("Q=1: strings in -> survivors out %.1f us in one call (bsg_query) vs %.1f us through hash + batch_create + probe; prepared batch: %.1f us per "
        "synchronous query in one dispatch (kernel %.1f us; %.1f us with a spin wait); two kernels + copy: %.1f us gathered (kernels %.1f + %.1f us), %.1f us streamed"
        % (res["end_to_end_bsg_query"]["latency_us_median"], res["end_to_end_three_calls"]["latency_us_median"],
           o["latency_us_median"], o["k_probe_direct_us"], res["one_dispatch_spin_wait"]["latency_us_median"], g["latency_us_median"],
           g_us, g.get("k_eval_programs_us", 0.0), res["stream"]["latency_us_median"]))
    return {"workload": "Q = 1: And(FT(level,error), FT(service,payment), FT(nested.region,region-3)) x %d blocks, survivors to host" % B,
            "survivors": int(sum(bin(int(x)).count("1") for x in first.ravel())), **res}


def concurrent_queries_leg(ctx, arenas, B, exprs, got, log, seconds=0.4):
    """The Go surface's call pattern: T host threads, each calling bsg_query with ONE query (a 3-term And(FieldToken) of the C2 batch)
    against 1 or 10 arenas — what the reference's file workers do (query_exec.go:303-357, 427-431: a goroutine per candidate file,
    several Query() calls at once).  Native threads (tools/native/conc_driver.cpp; Python threads would measure the interpreter
    lock).  Twice: with every call going alone (bsg_set_lab key 12 = 0: one k_query_direct dispatch per call, serialised on the
    device's stream — the round-4 behaviour) and with the combiner on (calls that meet share dispatches: a hot arena streamed once for
    all its callers, everything else one k_query_jobs dispatch).  `same arena`: every call names C2's arena; `12 arenas`: the calls
    rotate over 12 address-distinct replicas of it (distinct files).  Every result of every call is compared with the batch probe's
    rows inside the driver.  cpu_us_per_call = processor time of the whole process per call: the box's cgroup quota (cpu.max, quoted
    below) is what bounds the combined rate once hundreds of callers sleep and wake per call."""
    from bloomsearch_amd import conc
    nq = min(256, len(exprs))
    expected = np.ascontiguousarray(got[:nq])
    n_ar = min(len(arenas), 12)
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        cpus_quota = None if quota[0] == "max" else float(quota[0]) / float(quota[1])
    except Exception:  # noqa: BLE001
        cpus_quota = None
    res = {"queries": "%d distinct 3-term And(FieldToken) queries of the C2 batch, one per call" % nq, "seconds_per_point": seconds,
           "host_cpus": os.cpu_count(), "cgroup_cpu_quota_cpus": cpus_quota,
           "check": "every call's survivors compared with the batch probe's rows (bit-exact) inside the driver", "points": []}
    for apc, pool, name in ((1, arenas[:1], "same arena"), (1, arenas[:n_ar], "%d arenas" % n_ar), (10, arenas[:n_ar], "%d arenas" % n_ar)):
        if apc > len(pool):
            continue
        for T in (1, 16, 64, 256):
            row = {"threads": T, "arenas_per_call": apc, "arena_pool": name}
            for mode, mname in ((0, "alone"), (1, "combined")):
                ctx.set_lab(12, mode)
                ctx.query_stats(reset=True)
                r = conc.run(ctx, exprs[:nq], pool, B, expected, n_threads=T, seconds=seconds, arenas_per_call=apc)
                st = ctx.query_stats()
                if r["mismatches"] or r["errors"]:
                    sys.exit("concurrent_queries: %d mismatches, %d errors at T=%d, %d arenas per call, mode %s" % (r["mismatches"], r["errors"], T, apc, mname))
                row[mname] = {"queries_per_s": r["queries_per_s"], "probes_per_s": r["queries_per_s"] * apc * B * 3, "p50_us": r["p50_us"], "p99_us": r["p99_us"],
                              "calls": r["calls"], "cpu_us_per_call": r["cpu_us_per_call"], "cpus_busy": r["cpus_busy"]}
                if mode:
                    cyc = max(st["cycles"] - st["solo_calls"], 1)
                    row[mname].update({"cycles": st["cycles"], "calls_per_cycle": st["cycle_calls"] / max(st["cycles"], 1), "solo_cycles": st["solo_calls"],
                                       "max_calls_per_cycle": st["max_calls_per_cycle"], "dispatches_per_combined_cycle": st["dispatches"] / cyc,
                                       "hot_arenas_per_combined_cycle": st["hot_arenas"] / cyc,
                                       "collector_us_per_combined_cycle": {k[3:]: st[k] / cyc / 1e3 for k in ("ns_prepare", "ns_enqueue", "ns_wait", "ns_deal")}})
            row["speedup"] = row["combined"]["queries_per_s"] / max(row["alone"]["queries_per_s"], 1e-9)
            res["points"].append(row)
            log("concurrent queries: T=%3d x %2d arena(s) per call (%s): alone %.3g q/s (p50 %.0f us, p99 %.0f us, %.1f us cpu/call), combined %.3g q/s "
                "(p50 %.0f us, p99 %.0f us, %.1f us cpu/call, %.1f calls per cycle) = %.1fx"
                % (T, apc, name, row["alone"]["queries_per_s"], row["alone"]["p50_us"], row["alone"]["p99_us"], row["alone"]["cpu_us_per_call"],
                   row["combined"]["queries_per_s"], row["combined"]["p50_us"], row["combined"]["p99_us"], row["combined"]["cpu_us_per_call"],
                   row["combined"]["calls_per_cycle"], row["speedup"]))
    ctx.set_lab(12, 1)
    return res


def big_filter_leg(ctx, args, log):
    """Block filters BEYOND the LDS budget (VERDICT r03 missing 4): the reference's defaults (10 000 rows / 10 MiB per block,
    engine.go:127-128) give ~1 MB token filters once a row holds ~60 distinct tokens — 8.3 Mbit, seven times what a workgroup can
    stage (144 KiB).  Such a filter is never streamed into LDS: every (term, location) is a sector read from L2 / HBM, with the
    many-term mode's early termination (~3.3 of k = 10 locations per absent term).  64 blocks x 580 000 distinct 8-byte tokens
    (fill ~50 % as a right-sized filter has), a 29-term and a 4 054-term batch of single-Token queries; the kernel time is set
    against BOTH regimes' algorithmic bytes: every bitset once (what a windowed stream would move) and k x 8 B per
    (block, term) probe (what the gathers need)."""
    from bloomsearch_amd import _lib
    from bloomsearch_amd._lib import DESC_DTYPE
    from bloomsearch_amd.gpu import estimate_parameters
    from oracle import oracle as O
    n_blocks, per_block = 64, 580_000
    rng = np.random.default_rng(20260927)
    m, k = estimate_parameters(per_block, args.fpr)
    nw = (m + 63) // 64
    stride = (nw + 15) // 16 * 16
    t0 = time.time()
    toks = rng.integers(1, 1 << 62, size=n_blocks * per_block, dtype=np.uint64)          # 8 raw bytes per token (distinct with overwhelming odds)
    blob = toks.view(np.uint8)
    off = (np.arange(n_blocks * per_block + 1, dtype=np.uint64) * 8).astype(np.uint32)
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    fstart = [0]
    for b in range(n_blocks):
        fstart.append(b * per_block)                    # field: absent
        desc[b * 3 + 1] = (b * stride, m, k, 0)
        fstart += [(b + 1) * per_block, (b + 1) * per_block]
    words = ctx.build(blob, off, np.asarray(fstart, dtype=np.uint32), desc, n_blocks * stride)
    R = 4
    arenas = [ctx.arena_load(words, desc) for _ in range(R)]
    t_setup = time.time() - t0
    res = {"workload": "%d blocks x %d distinct tokens -> token filters of m = %d bits (%.2f MB, k = %d): %.1fx the LDS staging budget"
                       % (n_blocks, per_block, m, nw * 8 / 1e6, k, nw * 8 / (144 * 1024)),
           "bitset_bytes_per_arena": int(n_blocks * nw * 8)}
    for name, n_terms in (("few_terms", 29), ("many_terms", 4054)):
        n_present = max(1, n_terms // 10)
        present = toks[rng.integers(0, len(toks), size=n_present)]
        absent = rng.integers(1 << 62, 1 << 63, size=n_terms - n_present, dtype=np.uint64)
        tv = np.concatenate([present, absent])
        terms = np.zeros(n_terms, dtype=_lib.TERM_DTYPE)
        terms["h"] = ctx.hash_entries(tv.view(np.uint8), (np.arange(n_terms + 1, dtype=np.uint64) * 8).astype(np.uint32))
        terms["kind"] = 1
        ops = np.asarray([_lib.op(_lib.OP_TERM, i) for i in range(n_terms)], dtype=np.uint32)
        poff = np.arange(n_terms + 1, dtype=np.uint32)
        bid = ctx.batch_create(terms, ops, poff)
        got = ctx.probe_batch(arenas[0], bid, n_terms, n_blocks)
        if not args.no_check:
            want = O.probe_batch(words, desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
            if not np.array_equal(got, want):
                sys.exit("big filters (%s): survivors differ from the oracle" % name)
        ids = np.ascontiguousarray([arenas[i % R] for i in range(8)], dtype=np.uint64)
        for _ in range(3):
            ctx.probe_many(ids, bid, _lib.PROBE_ASYNC | _lib.PROBE_NOFUSE)
        ctx.sync()
        ctx.timing_read(reset=True)
        n = 10
        for _ in range(n):
            ctx.probe_many(ids, bid, _lib.PROBE_ASYNC | _lib.PROBE_NOFUSE | _lib.PROBE_TIMED)
        ctx.sync()
        tm = ctx.timing_read()
        ms = tm.ms_terms_kernel / max(tm.n_probes, 1)
        stream_bytes = len(ids) * n_blocks * nw * 8
        gather_bytes = len(ids) * n_blocks * n_terms * k * 8
        survive = sum(bin(int(x)).count("1") for x in got.ravel()) / (n_terms * n_blocks)
        res[name] = {"terms": n_terms, "arenas_per_launch": len(ids), "kernel": "k_probe_terms_many" if n_terms > 128 else "k_probe_terms",
                     "kernel_ms": ms, "pairs_surviving": survive,
                     "stream_regime": {"algorithmic_bytes": stream_bytes, "achieved": stream_bytes / ms / 1e6, "frac": stream_bytes / ms / 1e6 / HBM_PEAK_GBPS},
                     "gather_regime": {"algorithmic_bytes": gather_bytes, "achieved": gather_bytes / ms / 1e6, "frac": gather_bytes / ms / 1e6 / HBM_PEAK_GBPS,
                                       "note": "k x 8 B per (block, term); 8-byte words out of 64-byte sectors: 12.5% of peak is this regime's ceiling"},
                     "probes_per_s": len(ids) * n_blocks * n_terms / (ms * 1e-3)}
        ctx.batch_free(bid)
        log("big filters, %d terms: %.1f us per %d arenas of %d x %.2f MB = %.0f GB/s of bitsets (%.3f of peak if they were streamed), %.3g probes/s"
            % (n_terms, ms * 1e3, len(ids), n_blocks, nw * 8 / 1e6, res[name]["stream_regime"]["achieved"], res[name]["stream_regime"]["frac"],
               res[name]["probes_per_s"]))
    for a in arenas:
        ctx.arena_free(a)
    res["setup_s"] = t_setup
    res["check"] = "survivors of both batches bit-exact vs the oracle"
    return res


def c4_leg(ctx, args, rank, world, workers, log, barrier=None, headline=False):
    """BASELINE configs[3] / SURVEY C4: 100 M rows / 10 000 blocks as 10 files of 1 000 blocks, block b on rank b % N
    (strong scaling: the total is fixed), Q = 4096 8-term Or(FieldToken) queries.  One step probes the whole set once;
    every rank probes the blocks it holds of every file with bsg_probe_many (one arena per file)."""
    import torch
    from bloomsearch_amd import _lib, query as Q, synth
    from bloomsearch_amd.arena import plan_blocks
    n_files, per_file, rows, NQ = args.c4_files, args.c4_blocks_per_file, args.rows_per_block, args.queries
    total_blocks = n_files * per_file
    exprs = synth.make_queries(NQ, "c4", seed=4321)
    cb = Q.compile_queries(exprs)
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=_lib.TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    bid = ctx.batch_create(terms, ops, poff)
    t0 = time.time()
    files, ft_bytes, local_blocks = [], 0, []
    from oracle import oracle as O
    ok = True
    for f in range(n_files):
        gids = np.arange(f * per_file, (f + 1) * per_file, dtype=np.int64)
        gids = gids[gids % world == rank]
        blocks = generate_blocks(gids, rows, 0xB100F5EA4C4, workers)
        plan = plan_blocks(blocks, args.fpr)
        del blocks
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        files.append((words, plan.desc))
        local_blocks.append(len(gids))
        ft_bytes += int(sum((int(m) + 63) // 64 * 8 for m in plan.desc["m"][2::3]))
        del plan
    R = max(2, int(np.ceil(2 * 256 * 2 ** 20 / max(ft_bytes, 1))))
    reps = [[ctx.arena_load(w, d) for (w, d) in files] for _ in range(R)]
    log("c4: %d files x %d blocks (%d held by this rank, %.1f MB of FT bitsets per step), %d replicas, %d distinct terms; setup %.1fs"
        % (n_files, per_file, sum(local_blocks), ft_bytes / 1e6, R, len(terms), time.time() - t0))
    G = [(nb + 63) // 64 for nb in local_blocks]
    words_per_step = NQ * sum(G)
    # correctness: this rank's shard of every file against the oracle (first queries), outside the timed region
    got = ctx.probe_many(reps[0], bid, 0, NQ, local_blocks)
    nchk = min(48, NQ)
    sel = np.sort(np.random.default_rng(4321 + rank).choice(NQ, size=nchk, replace=False))     # a random sample, a different one per rank
    if not args.no_check:
        for f in range(n_files):
            if local_blocks[f] == 0:
                continue
            w, d = files[f]
            want = O.survivors_tree(w, d.view(O.DESC_DTYPE), [exprs[int(i)] for i in sel])          # the tree-walking evaluator
            if not np.array_equal(got[f][sel], want):
                ok = False
    if world > 1:
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], device=COLL_DEVICE())
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if not ok:
        sys.exit("c4: survivor sets differ from the oracle — refusing to report")
    files = None
    pr = Prober(ctx, bid, world, log, barrier)
    # as the line's headline (N > 1) the leg times EXACTLY --steps steps after --warmup untimed ones, like the contract says;
    # as a side leg (N = 1) it is bounded
    steps = max(1, args.steps) if headline else max(4, min(args.steps, 60))
    per_call = max(1, args.group // max(n_files, 1))   # steps handed to one bsg_probe_many call (<= --group arenas per dispatch)
    make = lambda i: reps[i % R]
    c4_warm = max(0, args.warmup) if headline else max(2, min(args.warmup, 8))
    # untimed setup: one call of the timed region's shape sizes the library's verdict / survivor scratch (the warmup steps may be
    # fewer than one call covers, and a first call that grows the scratch calls hipMalloc inside the region: +8 us per step, measured)
    pr.run(pr.plan([make(c4_warm + i) for i in range(min(per_call, steps))], per_call), 0)     # (the timed region's own first call: the steady state of a host that probes the same files again)
    ctx.sync()
    if args.events_in_headline:
        dt, tm = pr.measure(make, steps, c4_warm, per_call)
        c4_clock = dict(pr.last_closing)
        dt_ev = dt
    else:                                                # bare first (the number), then the same steps with dispatch timestamps (the kernel durations)
        dt, _ = pr.measure(make, steps, c4_warm, per_call, timed=False)
        c4_clock = dict(pr.last_closing)
        dt_ev, tm = pr.measure(make, steps, 2, per_call)
    c4_kernels = kernel_stats(tm, len(terms))          # 77 distinct terms: the few-term kernel
    c4_dom = dominant_kernel(tm)
    probes = NQ * total_blocks * 8
    res = {"workload": "C4: %d rows/block x %d blocks in %d files, block b on rank b %% %d, Q=%d 8-term Or(FieldToken), %d distinct terms; "
                       "%d address-distinct replicas rotated per step" % (rows, total_blocks, n_files, world, NQ, len(terms), R),
           "scaling": "strong", "n_gpus": world, "blocks_total": total_blocks, "steps": steps, "ms_per_step": dt / steps * 1e3, "value": probes * steps / dt,
           "ms_per_step_with_dispatch_timestamps": dt_ev / steps * 1e3, "clock": dict(c4_clock, note=CLOCK_NOTE),
           "unit": "probes/s", "probes_per_step": probes, "stream_bytes_per_step_per_gpu": ft_bytes,
           "kernels": c4_kernels, "dominant_kernel": c4_dom, "warmup": c4_warm,
           "blocks_held_by_rank0": int(sum(local_blocks)),
           "check": "every rank's shard of every file bit-exact vs the tree-walking oracle on %d randomly chosen queries" % nchk}
    if world > 1:
        # every rank's own kernel time and launch count (rank order): a straggler GPU shows here, not in the max-over-ranks wall time
        import torch.distributed as dist
        kd = c4_kernels.get(c4_dom) or {}
        mine = torch.tensor([kd.get("kernel_ms", 0.0), float(kd.get("samples", 0)), tm.ms_eval_kernel / max(tm.n_eval, 1),
                             float(sum(local_blocks))], dtype=torch.float64, device=COLL_DEVICE())
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        res["per_rank"] = [{"rank": r, "kernel": c4_dom, "kernel_ms": float(t[0]), "launches": int(t[1]), "k_eval_programs_ms": float(t[2]),
                            "blocks": int(t[3])} for r, t in enumerate(allr)]
    # the same steps with the host-side gather inside the timed region
    slot_words = words_per_step * per_call
    n_slots = 2
    hdr_per_step = NQ * n_files
    sh = SharedHost(ctx, slot_words * 8 * n_slots + hdr_per_step * per_call * 4 * n_slots, rank, world, "c4")
    ring = sh.mine[: slot_words * n_slots]
    ro = Ring(ring)          # successive calls write successive slots of the shared segment (only the last ones survive)
    dt2, _ = pr.measure(make, steps, 2, per_call, timed=False, out=ro, words_per_step=words_per_step)
    res["host_gather"] = {"ms_per_step": dt2 / steps * 1e3, "value": probes * steps / dt2,
                          "survivor_bytes_per_step_per_gpu": words_per_step * 8, "page_locked": sh.registered,
                          "note": "every rank's survivors DMA-ed into one shared page-locked host segment that rank 0 reads "
                                  "(copy stream, overlapped with the next dispatch)"}
    dense_last = ro.last
    if sh.registered:
        # the same steps delivered as survivor ROWS: a 4-byte header per (file, query) + block ids / words only where the row needs
        # them (bsg_probe_many_rows), written by the device into the same segment.  An 8-term Or keeps every block of a file for
        # nearly every query: those rows are a header and nothing else.
        from bloomsearch_amd.gpu import rows_to_dense
        hdr_ring = sh.mine[slot_words * n_slots:].view(np.uint32)[: hdr_per_step * per_call * n_slots]
        rr, hr = Ring(ring), Ring(hdr_ring)
        dt3, _ = pr.measure(make, steps, 2, per_call, timed=False, out=rr, words_per_step=words_per_step, hdr=hr, hdr_per_step=hdr_per_step)
        if C.ROWS_PACKED:       # byte headers (tag << 6 | a LIST row's count) at the front of the call's header region
            h_all = sh.part(rank)[slot_words * n_slots:].view(np.uint8)[hr.last * 4: hr.last * 4 + hdr_per_step]
            tag_all, cnt = h_all >> 6, (h_all & 63).astype(np.int64)
        else:
            h_all = sh.part(rank)[slot_words * n_slots:].view(np.uint32)[hr.last: hr.last + hdr_per_step]
            tag_all, cnt = h_all >> 30, (h_all & np.uint32(0x3FFFFFFF)).astype(np.int64)
        tags = np.bincount(tag_all, minlength=4)
        payload = int(4 * cnt[tag_all == 2].sum() + 8 * sum(G[f] * int((tag_all[f * NQ: (f + 1) * NQ] == 3).sum()) for f in range(n_files)))
        o = 0
        for f in range(n_files):                                   # the rows of the last call's first step expand to the direct probe's bitsets
            if local_blocks[f] and not args.no_check:
                back = rows_to_dense(h_all[f * NQ: (f + 1) * NQ], sh.part(rank)[rr.last + o: rr.last + o + NQ * G[f]], local_blocks[f], packed=C.ROWS_PACKED)
                if not np.array_equal(back, got[f]):
                    sys.exit("c4: survivor rows of file %d do not expand to the direct probe's bitsets" % f)
            o += NQ * G[f]
        res["host_gather"]["rows"] = {"api": "bsg_probe_many_rows" + (" (BSG_PROBE_ROWS_PACKED)" if C.ROWS_PACKED else ""), "ms_per_step": dt3 / steps * 1e3, "value": probes * steps / dt3,
                                      "bytes_per_step_per_gpu": int((1 if C.ROWS_PACKED else 4) * hdr_per_step + payload),
                                      "rows_by_tag_none_all_list_dense": [int(x) for x in tags],
                                      "vs_device_resident": dt3 / dt,
                                      "note": "header + ids / words where needed, written by the device into the page-locked segment; "
                                              "vs_device_resident = this step time over the step time with the survivors left on the device"}
        # the dense pass is checked below: run it once more so that the segment holds bitsets again
        ro = Ring(ring)
        pr.measure(make, min(steps, per_call), 0, per_call, timed=False, out=ro, words_per_step=words_per_step)
        dense_last = ro.last
    if rank == 0:
        # what the consumer does: rank 0 reads every rank's slice of the segment and interleaves global block order
        from bloomsearch_amd import parallel as P
        mine = sh.part(0)[dense_last: dense_last + words_per_step]
        o = 0
        for f in range(n_files):
            if not np.array_equal(mine[o: o + NQ * G[f]].reshape(NQ, G[f]), got[f]):
                sys.exit("c4: survivors delivered to the shared host segment differ from the direct probe")
            o += NQ * G[f]
        if per_file % world == 0 and n_files > 0:
            parts = [sh.part(r)[dense_last: dense_last + NQ * G[0]].reshape(NQ, G[0]) for r in range(world)]
            glob = P.interleave_survivors(parts, per_file)
            if not np.array_equal(np.ascontiguousarray(glob[:, : 1]) & np.uint64(1), got[0][:, :1] & np.uint64(1)):
                sys.exit("c4: global block 0 of file 0 (held by rank 0) changed in the interleave")
            res["host_gather"]["rank0_view"] = "file 0: %d ranks' bitsets interleaved into [%d][%d] global words" % (world, NQ, glob.shape[1])
    sh.close()
    log("c4: %.1f us/step = %.3g probes/s device-resident; %.1f us/step = %.3g probes/s with the host gather (dense bitsets); rows: %s"
        % (dt / steps * 1e6, res["value"], dt2 / steps * 1e6, res["host_gather"]["value"],
           ("%.1f us/step" % (res["host_gather"]["rows"]["ms_per_step"] * 1e3)) if "rows" in res["host_gather"] else "-"))
    for rep in reps:
        for a in rep:
            ctx.arena_free(a)
    ctx.batch_free(bid)
    return res


def traffic_from_profiles(dom, k, args, B):
    """HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes of this same command (tools/profile.sh),
    corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 for wide coalesced reads on gfx950, + WRITE_SIZE), committed under
    profiles/ — the builder's number from an earlier run of the same shape, labelled as such in traffic_source."""
    try:
        tname = next(n for n in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        rec = json.load(open(os.path.join(ROOT, "profiles", tname)))
        per_arena = rec[dom]["hbm_bytes_corrected_per_arena"]
        if args.workload == "c2" and B == 1000 and k:
            return (per_arena * k["arenas_per_launch"],
                    "profiles/%s: rocprofv3 --pmc passes of this command, per 1 000-block arena x arenas per launch" % tname)
    except Exception:  # noqa: BLE001
        pass
    return None, None


def valu_issue_from_profiles(kernel, k, B, n_kinds=1):
    """The VALU-issue fraction of a kernel whose bound is not HBM (the many-term probe): VALU instructions per wave from the committed SQ
    counter pass (tools/profile_pmc.sh -> profiles/rNN_needle_pmc.txt) x 4 cycles of a SIMD per wave64 instruction x the waves of one
    launch (blocks x kinds x 8 waves x arenas), over the SIMD-cycles of this run's measured kernel time (1 024 SIMDs at 2.4 GHz)."""
    if not k or not k.get("kernel_ms"):
        return None
    try:
        tname = next(n for n in ("r05_needle_pmc.txt", "r04_needle_pmc.txt", "r03_needle_pmc.txt") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        per_wave = None
        with open(os.path.join(ROOT, "profiles", tname)) as f:
            inside = False
            for ln in f:
                if ln.startswith("=="):
                    inside = ln.split()[1].rstrip(",") == kernel
                elif inside and ln.split()[:1] == ["SQ_INSTS_VALU"] and "per wave" in ln:
                    per_wave = float(ln.split()[-1])
        if per_wave is None:
            return None
        waves = k["arenas_per_launch"] * B * n_kinds * 8
        simd_cycles = k["kernel_ms"] * 1e-3 * 2.4e9 * 1024
        return {"valu_issue_frac": per_wave * 4 * waves / simd_cycles, "valu_insts_per_wave": per_wave, "waves_per_launch": waves,
                "source": "profiles/%s (SQ_INSTS_VALU per wave) x 4 SIMD cycles x waves, over this run's kernel time x 1 024 SIMDs x 2.4 GHz" % tname}
    except Exception:  # noqa: BLE001
        return None


def scaled_leg(ctx, pr, args, plan, words, B, NQ, ft_bytes, n_terms, terms_per_query, log):
    """C2' (SURVEY 8d): the same arena replicated x S inside ONE arena, one launch — steady-state streaming bandwidth next to the
    launch-latency-bound 35 MB case (N = 1 only)."""
    S = args.scaled
    t0 = time.time()
    big = ctx.arena_load(words, np.tile(plan.desc, S))       # S address-distinct copies of every filter
    log("scaled arena: %d blocks (%.2f GB of FT bitsets per launch) loaded in %.1fs" % (B * S, ft_bytes * S / 1e9, time.time() - t0))
    s_steps = max(4, min(args.steps, 20))
    ctx.set_probe_group(1)
    s_elapsed, s_tm = pr.measure(lambda i: [big], s_steps, 2, 1, nofuse=True)
    ctx.set_probe_group(args.group)
    s_bytes = s_tm.stream_bytes / max(s_tm.n_probes, 1) + 33 * n_terms
    s_ms = s_tm.ms_terms_kernel / max(s_tm.n_probes, 1)
    scaled = {"blocks": B * S, "steps": s_steps, "ms_per_step": s_elapsed / s_steps * 1e3,
              "value": NQ * B * S * terms_per_query * s_steps / s_elapsed, "kernel_ms": s_ms,
              "eval_kernel_ms": s_tm.ms_eval_kernel / max(s_tm.n_eval, 1),
              "algorithmic_bytes_per_launch": s_bytes, "achieved": s_bytes / (s_ms * 1e-3) / 1e9,
              "frac": s_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    ctx.arena_free(big)
    return scaled


def decode_leg(ctx, plan, words, B, log):
    """a8 (file_format.go:343-448) on the device, both directions: the same 1 000 blocks built and serialised as on-disk filter sections
    (big-endian words + CRC32C) by bsg_build_sections, then uploaded as bytes and decoded by k_decode_sections — in one launch, as the
    call runs by default (four launches behind the copy) and through the cursor-shaped API (a9).  A sample of sections is compared with
    the host codec fed from the words of bsg_build."""
    from bloomsearch_amd import host as Hst
    t0 = time.time()
    secs = ctx.build_sections(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    t_enc = time.time() - t0
    enc_ms = ctx.last_encode_ms()
    for b in (0, B // 2, B - 1):
        fl = []
        for c in range(3):
            d = plan.desc[b * 3 + c]
            nw = (int(d["m"]) + 63) // 64
            fl.append((int(d["m"]), int(d["k"]), words[int(d["word_off"]): int(d["word_off"]) + nw]))
        if Hst.section_encode(fl) != secs[b]:
            sys.exit("device-encoded section %d differs from the host codec" % b)
    sec_bytes = sum(len(x) for x in secs)
    # the kernel by itself: one launch over all sections, after the whole copy (lab key 4 = 1) ...
    ctx.set_lab(4, 1)
    sid, st = ctx.arena_load_sections(secs)
    dec_one_ms = ctx.last_kernel_ms()[2]
    if st.any():
        sys.exit("device section decode reported failures on clean sections")
    ctx.arena_free(sid)
    # ... and as the call runs by default: four launches, each behind its quarter of the copy
    ctx.set_lab(4, 4)
    t1 = time.time()
    sid, st = ctx.arena_load_sections(secs)
    t2 = time.time()
    dec_ms = ctx.last_kernel_ms()[2]
    if st.any():
        sys.exit("device section decode reported failures on clean sections")
    ctx.arena_free(sid)
    # a9: the same region through the cursor-shaped API, 4 MiB at a time as blockFilterCursor reads it
    # (file_format.go:618): the copy of chunk i + 1 overlaps the decode of chunk i
    blob_secs = b"".join(secs)
    offs = np.zeros(len(secs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in secs])
    t3 = time.time()
    stream = ctx.arena_stream_begin(offs[:-1], offs[1:])
    for o in range(0, len(blob_secs), 4 << 20):
        ctx.arena_stream_append(stream, o, blob_secs[o: o + (4 << 20)])
    sid2, st2 = ctx.arena_stream_finish(stream, len(secs))
    t4 = time.time()
    stream_dec_ms = ctx.last_kernel_ms()[2]
    if st2.any():
        sys.exit("streamed section decode reported failures on clean sections")
    ctx.arena_free(sid2)
    decode = {"kernel": "k_decode_sections", "kernel_ms": dec_one_ms, "section_bytes": sec_bytes,
              "algorithmic_bytes": 2 * sec_bytes, "achieved": 2 * sec_bytes / max(dec_one_ms, 1e-6) / 1e6, "unit": "GB/s",
              "note": "CRC32C + BE->LE decode of %d filter sections on the device in one launch; bytes = sections read + words written" % B,
              "pieces": {"launches": 4, "kernel_ms_sum": dec_ms, "end_to_end_s_incl_h2d": t2 - t1,
                         "note": "bsg_arena_load_sections as it runs by default: the decode of each quarter starts behind its part of the copy "
                                 "(round 4: a section is cut into 16 KB slices, one workgroup each, so a quarter's ~250 sections are ~1 100 workgroups)"},
              "end_to_end_s_incl_h2d": t2 - t1,
              "stream": {"api": "bsg_arena_stream_begin / append (4 MiB chunks) / finish", "chunks": (len(blob_secs) + (4 << 20) - 1) // (4 << 20),
                         "end_to_end_s_incl_h2d": t4 - t3, "decode_kernels_ms_sum": stream_dec_ms,
                         "note": "sections are parsed (flags, lengths, m, k), CRC-checked and decoded on the device as their last byte lands"},
              "encode": {"kernels": "k_encode_payload + k_crc_sections", "kernel_ms": enc_ms,
                         "achieved": 3 * sec_bytes / max(enc_ms, 1e-6) / 1e6, "unit": "GB/s",
                         "note": "LE->BE + framing + CRC32C of the same sections on the device (bsg_build_sections); bytes = "
                                 "words read + sections written + sections re-read by the checksum pass",
                         "build_and_encode_end_to_end_s_incl_copies": t_enc}}
    log("device section codec: %.1f MB of sections encoded in %.1f us (%.0f GB/s), decoded in %.1f us by one launch (%.0f GB/s; "
        "%.1f us as four launches behind the copy), %.3fs incl. H2D"
        % (sec_bytes / 1e6, enc_ms * 1e3, decode["encode"]["achieved"], dec_one_ms * 1e3, decode["achieved"], dec_ms * 1e3, t2 - t1))
    return decode

