#!/usr/bin/env python3
"""bench.py — block-bloom probe throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the probe hot path over one batch of synthetic input: Q = 4096 three-term
And(FieldToken, FieldToken, FieldToken) queries against the filters of one 10M-row / 1 000-block file set
(BASELINE configs[1]; 10 000 rows per block, fpr 0.001) = Q x B x 3 (block, term) probes.  Filter arena, hashed term
table and compiled programs are resident in HBM before the timed region; the survivor bitsets stay on the device
(`value`); the same steps with the survivors delivered to host memory are `value_survivors_delivered_to_host`.
Step i probes arena replica i % R (R x streamed bytes >= 2 x the 256 MiB Infinity Cache), and consecutive steps go
to bsg_probe_many together (one dispatch covers up to --group arenas).

stdout carries ONE compact JSON line (< 4 KB, strict JSON): metric, value, roofline, cpu_baseline, clock, and the
one-number summaries of BASELINE's other configs (build = configs[2], c4 = configs[3], or_reduce = configs[4]).
Every leg's full object goes to bench_legs.json beside this script (and a digest of each to stderr).

N > 1 (torchrun, one process per GPU): the headline is BASELINE configs[3] (C4: 10 000 blocks in total, block b on rank
b % N, 8-term Or batch, strong scaling); the weak-scaling C2 run moves to bench_legs.json["c2_weak"].  The C4 curve over
N reads: this line's `value` at N > 1, `c4.value` of the N = 1 line.

The harness lives in benchlib/ (common.py: generator pool, shared host segment, Prober; legs.py: every other leg).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib import common as C      # noqa: E402  (numpy only: torch / HIP are imported after the generator pool has forked)
from benchlib.common import HBM_PEAK_GBPS, Prober, Ring, SharedHost, generate_blocks, kernel_stats, dominant_kernel, merge_timing  # noqa: E402

LINE_LIMIT = 4096                     # the driver's parser gave up on a 25 KB line in round 5: the headline stays under this, checked before printing


def go_reference_baseline(rows, n_blocks_sample, log):
    """The reference's own Go CPU path (go/overlay/bench_reference_test.go::BenchmarkReferenceProbeLoop) when this box has
    a Go toolchain and BLOOMSEARCH_REFERENCE points at a checkout whose modules resolve; None otherwise (the usual case:
    neither the build image nor the GPU box has Go, and /root/reference does not travel)."""
    import re
    import shutil
    import subprocess
    import tempfile
    go, ref = shutil.which("go"), os.environ.get("BLOOMSEARCH_REFERENCE")
    if not go or not ref or not os.path.isdir(ref):
        log("cpu_baseline: no Go toolchain / BLOOMSEARCH_REFERENCE here (go=%s): timing the oracle's restatement instead" % go)
        return None
    from bloomsearch_amd import synth
    with tempfile.NamedTemporaryFile("wb", suffix=".ndjson", delete=False) as f:
        for b in range(n_blocks_sample):
            f.write(b"\n".join(synth.rows_json(b * rows, rows)) + b"\n")
        path = f.name
    try:
        out = subprocess.run([os.path.join(ROOT, "go", "run_bench_reference.sh"), ref, path, str(rows)],
                             capture_output=True, text=True, timeout=900)
        m = re.search(r"BenchmarkReferenceProbeLoop\S*\s+\d+\s+.*?([0-9.e+]+) cores\s+([0-9.e+]+) probes/s", out.stdout)
        if out.returncode != 0 or not m:
            log("cpu_baseline: go harness failed (rc %d): %s" % (out.returncode, (out.stderr or out.stdout)[-400:]))
            return None
        return {"value": float(m.group(2)), "unit": "probes/s", "cores": int(float(m.group(1))), "kind": "reference",
                "sample": "BenchmarkReferenceProbeLoop: 256 C2-shaped queries x %d blocks of %d rows (parseFilterSection + "
                          "evaluateBloomFilters per (query, block)), real bloom/v3" % (n_blocks_sample, rows)}
    finally:
        os.unlink(path)




def cpu_baseline(words, desc, cb, ops, poff, n_blocks, budget_s, log, terms_per_query=3):
    """The reference's probe loop as the oracle restates it (parse section incl. CRC32C + BE decode, then short-circuit
    TestString with per-call re-hash: file_format.go:393-448 + query_exec.go:75-159 per (query, block)), on a bounded sample.
    Threads = the CPUs this process may really use (affinity mask and cgroup quota: query_exec.go:245-357 sizes its worker pool
    the same way, MaxQueryConcurrency ~ cores) — `cores` in the line is that number, `logical_cpus` what the box shows."""
    from oracle import oracle as O
    cores, logical = C.effective_cpus()
    secs, sec_off, max_words = [], [0], 0
    for b in range(n_blocks):
        fl = []
        for c in range(3):
            d = desc[b * 3 + c]
            if d["m"] == 0:
                fl.append(None)
                continue
            nw = (int(d["m"]) + 63) // 64
            max_words = max(max_words, nw)
            fl.append(O.Filter(int(d["m"]), int(d["k"]), words[int(d["word_off"]): int(d["word_off"]) + nw]))
        s = O.encode_filter_section(fl)
        secs.append(s)
        sec_off.append(sec_off[-1] + len(s))
    sections = b"".join(secs)
    sec_off = np.asarray(sec_off, dtype=np.uint64)
    kinds = np.asarray(cb.term_kinds, dtype=np.uint32)

    def run(nq):
        t0 = time.perf_counter()
        out = O.probe_reference_style(sections, sec_off, max_words, cb.term_strings, kinds,
                                      ops[: poff[nq]], poff[: nq + 1], n_threads=cores)
        return time.perf_counter() - t0, out

    nq = min(cores, cb.n_queries)
    t, _ = run(nq)
    scale = max(1, int(budget_s / max(t, 1e-3)))
    nq2 = min(cb.n_queries, nq * scale)
    nq2 = max(cores, nq2 // cores * cores)
    t2, out = run(nq2)
    value = nq2 * n_blocks * terms_per_query / t2
    log("cpu_baseline: %d queries x %d blocks in %.2fs on %d threads (%d logical CPUs on the box)" % (nq2, n_blocks, t2, cores, logical))
    return {"value": value, "unit": "probes/s", "cores": cores, "logical_cpus": logical, "kind": "port",
            "sample": "%d of %d queries x %d blocks, %.1fs: oracle restatement of parseFilterSection + evaluateBloomFilters per "
                      "(query, block), threads = usable CPUs" % (nq2, cb.n_queries, n_blocks, t2)}, out, nq2


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no RANK / WORLD_SIZE in the environment): this process
    replaces itself with `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>` — one rank per GPU, the same command the driver's scaling tier uses; stdout (the one JSON
    line rank 0 prints) stays this process' stdout."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import socket
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus and os.environ.get("BSG_BENCH_SHARE_GPU") != "1":
        sys.exit("bench.py --gpus %d: %d GPU(s) visible (BSG_BENCH_SHARE_GPU=1 runs every rank on device 0 as a functional check "
                 "of the N > 1 host paths; its numbers mean nothing)" % (args.gpus, n_dev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: re-executing as %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)



def jsonable(x):
    """Strict JSON: numpy scalars -> Python numbers, NaN / inf -> null (json.dumps(..., allow_nan=False) then never raises)."""
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, np.generic):
        x = x.item()
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    return x


def sig(x, n=6):
    """n significant digits (the line is read by a parser and by people: 1854905447716.6006 says nothing 1.85491e12 does not)."""
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    if isinstance(x, float) and math.isfinite(x) and x != 0.0:
        return float("%.*g" % (n, x))
    return x


def pick(d, *keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def headline(h, legs):
    """The ONE line of stdout: the contract's keys + roofline + cpu_baseline + clock, and one small object per further BASELINE config."""
    line = dict(h)
    b = legs.get("build")
    if b:
        line["build"] = pick(b, "kernel", "kernel_ms", "first_call_ms", "calls", "algorithmic_bytes", "achieved", "frac", "entries_per_s")
    c4 = legs.get("c4")
    if c4 and h["n_gpus"] == 1:
        ck = (c4.get("kernels") or {}).get(c4.get("dominant_kernel")) or {}
        line["c4"] = dict(pick(c4, "value", "ms_per_step", "steps", "scaling"), blocks=c4.get("blocks_total"), kernel=c4.get("dominant_kernel"),
                          kernel_ms=ck.get("kernel_ms"), frac=ck.get("frac"),
                          value_survivors_delivered_to_host=((c4.get("host_gather") or {}).get("rows") or c4.get("host_gather") or {}).get("value"))
    s = legs.get("roofline_scaled")
    if s:
        line["roofline_scaled"] = pick(s, "blocks", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "achieved", "frac")
    o = legs.get("or_reduce")
    if o:
        line["or_reduce"] = pick(o, "kernel", "kernel_ms", "filters_this_rank", "achieved", "frac", "allreduce_ms", "allreduce_gbps_in_per_gpu",
                                 "n_ranks_seen_by_rccl", "allreduce_error")
    line["legs"] = "bench_legs.json: " + ", ".join(sorted(legs))
    text = json.dumps(sig(jsonable(line)), allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:          # never let an addition silently break the driver's parser again: shed the optional objects
        for k in ("legs", "or_reduce", "roofline_scaled", "c4", "build"):
            line.pop(k, None)
            text = json.dumps(sig(jsonable(line)), allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_legs(legs, log):
    path = os.path.join(ROOT, "bench_legs.json")
    try:
        with open(path, "w") as f:
            json.dump(jsonable(legs), f, indent=1, allow_nan=False)
            f.write("\n")
        log("every leg's full object: %s" % path)
    except OSError as exc:
        log("bench_legs.json not written (%r): the legs follow on stderr" % (exc,))
        print(json.dumps(jsonable(legs), allow_nan=False), file=sys.stderr, flush=True)


def closing_legs(emit, legs, ctx, args, rank, world, local_rank, plan, words, block_ids, rows, terms, ops, poff, got, or_reduce, or_state, log):
    """The legs that may not come back (one context over every GPU of the job; the RCCL exchange), each under a watchdog that still
    prints the line; then the line itself — exactly once."""
    import threading
    import torch.distributed as dist
    from benchlib import legs as LG
    # ---- one process, one context over every GPU of the job (rank 0, after every timed leg; the other ranks wait at the
    # barrier below).  BSG_BENCH_MULTI_CTX=n (lab): a context of n entries that all name this rank's GPU. ----
    if rank == 0:
        import torch
        n_lab = int(os.environ.get("BSG_BENCH_MULTI_CTX", "0"))
        ids = [local_rank] * n_lab if n_lab > 1 else (
            list(range(world)) if world > 1 and C.COLL_DEVICE() == "cuda" and torch.cuda.device_count() >= world else None)
        if ids:
            finished = threading.Event()
            mdc = legs["multi_device_context"] = {"devices": ids}      # filled in place, stage by stage

            def giving_up():
                if not finished.wait(float(os.environ.get("BSG_BENCH_MULTI_CTX_TIMEOUT", "240"))):
                    mdc["error"] = "stage %r gave no answer within the watchdog's time; the leg was abandoned (stages_done lists what did finish)" % mdc.get("stage_running")
                    emit()
                    os._exit(0)
            threading.Thread(target=giving_up, daemon=True).start()
            try:
                LG.multi_device_context_leg(ctx, ids, plan, words, block_ids, rows, 0xB100F5EA4C4, args.fpr, terms, ops, poff, got, log, res=mdc)
            except Exception as exc:  # noqa: BLE001 - reported in the legs
                mdc["error"] = "stage %r: %r" % (mdc.get("stage_running"), exc)
                log("multi-device context leg failed: %r" % (exc,))
            finished.set()
        elif world > 1:
            legs["multi_device_context"] = {"skipped": "%d device(s) visible to rank 0" % torch.cuda.device_count()}
    if world > 1:
        dist.barrier()
    if or_state is not None:
        # the RCCL leg last, under a watchdog: if a collective never returns, rank 0 still prints the line (with the error
        # noted) and every rank leaves — a hung collective cannot be cancelled from Python
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("BSG_BENCH_RCCL_TIMEOUT", "120"))):
                if rank == 0:
                    or_reduce["allreduce_error"] = "no answer within the watchdog's time; the leg was abandoned"
                    emit()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        LG.or_exchange_leg(ctx, or_reduce, or_state, world, log)
        done.set()
    emit()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=1000, help="blocks per GPU")
    ap.add_argument("--rows-per-block", type=int, default=10000)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--workload", default="c2", choices=["c2", "needle", "c4"],
                    help="c2: SURVEY 8d C2 And(FT(level), FT(service), FT(nested.region)); needle: third term is FT(user_id) (~4k distinct terms); c4: SURVEY C4's 8-term Or(FieldToken...)")
    ap.add_argument("--replicas", type=int, default=0, help="address-distinct arena replicas (0 = auto)")
    ap.add_argument("--fpr", type=float, default=0.001)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the device section-decode measurement")
    ap.add_argument("--or-union", type=int, default=200000,
                    help="distinct-entry count the fixed OR-reduce geometry is sized for (C5 leg); 0 = skip")
    ap.add_argument("--ingest-blocks", type=int, default=1000,
                    help="blocks of JSON rows pushed through the device ingest path (k_ingest_rows ...): 1000 = BASELINE configs[2]'s "
                         "10 M rows (2.5 GB of JSON); 0 = skip")
    ap.add_argument("--scaled", type=int, default=64,
                    help="also time C2': the arena replicated this many times inside one launch (0/1 = skip)")
    ap.add_argument("--group", type=int, default=256, help="arenas one probe dispatch may cover (bsg_set_probe_group; beyond 128 the arena records travel in device memory)")
    ap.add_argument("--samples", type=int, default=16, help="timestamped dispatches of each kernel beyond the timed region")
    ap.add_argument("--c4-files", type=int, default=10, help="files of the c4 leg (0 = skip the leg)")
    ap.add_argument("--c4-blocks-per-file", type=int, default=1000)
    ap.add_argument("--compact-rounds", type=int, default=-1, help="lab: compaction rounds of the many-term probe mode (bsg_set_lab key 1)")
    ap.add_argument("--fuse-limit", type=int, default=-1, help="lab: largest group whose evaluation rides in the next group's probe launch (bsg_set_fuse_limit)")
    ap.add_argument("--events-in-headline", action="store_true",
                    help="carry the per-dispatch HIP timestamps inside the headline region too (they cost ~17 us per 2-dispatch call: the default "
                         "times the K steps bare, then repeats them with the timestamps on and reports both)")
    ap.add_argument("--fold", type=int, default=-1, help="lab: evaluators per tile of k_probe_eval (bsg_set_lab key 11; 0 = two dispatches, the default)")
    ap.add_argument("--tail-split", type=int, default=-1, help="lab: percent of a single-group run evaluated on a second stream beside the rest's probe (bsg_set_lab key 19)")
    ap.add_argument("--no-q1", action="store_true", help="skip the Q = 1 latency leg")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the concurrent_queries leg (T host threads x bsg_query)")
    ap.add_argument("--no-big-filters", action="store_true", help="skip the leg with block filters beyond the LDS budget (~1 MB each)")
    ap.add_argument("--no-single", action="store_true", help="skip the one-arena-per-launch sampling pass")
    return ap.parse_args()


def main():
    args = parse_args()
    self_launch(args)

    # stdout carries exactly one JSON line: libraries that print banners through C stdio (librccl writes its version block
    # to stdout when a communicator is created, flushed at exit — i.e. AFTER the JSON) are pointed at stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d found itself alone after the self-launch (WORLD_SIZE=1)" % args.gpus)
        sys.exit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
    log = (lambda *a: print("[bench]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    usable_cpus, _ = C.effective_cpus()
    workers = min(32, max(1, usable_cpus // max(1, min(world, 8))))
    C.start_pool(workers)        # the generator's workers: forked before torch / HIP / RCCL exist here

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the probe path has no CPU fallback")
    # lab: BSG_BENCH_SHARE_GPU=1 runs every rank on device 0 with gloo collectives — a functional check of the N > 1 host
    # paths (sharding, shared-segment gather, rank-0 assembly) on a 1-GPU box; RCCL refuses two ranks on one device, so the
    # OR all-reduce leg is skipped there and the numbers mean nothing
    share_gpu = world > 1 and os.environ.get("BSG_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    barrier = C.HostBarrier(rank, world)

    from benchlib import legs as LG
    from bloomsearch_amd import _lib, query as Q, synth
    from bloomsearch_amd.arena import plan_blocks
    from bloomsearch_amd.gpu import Context

    ctx = Context((local_rank,))
    ctx.set_probe_group(args.group)
    for key, val in ((1, args.compact_rounds), (11, args.fold), (19, args.tail_split)):
        if val >= 0:
            ctx.set_lab(key, val)
    if args.fuse_limit >= 0:
        ctx.set_fuse_limit(args.fuse_limit)
    B, rows, NQ = args.blocks, args.rows_per_block, args.queries
    legs = {}                       # every leg's full object -> bench_legs.json (rank 0)

    # ---- untimed setup: this rank's shard = global blocks rank, rank + world, ... (round-robin) ----
    t0 = time.time()
    block_ids = np.arange(B, dtype=np.int64) * world + rank
    plan = plan_blocks(generate_blocks(block_ids, rows, 0xB100F5EA4C4, workers), args.fpr)
    log("generated %d blocks (%d entries, %.0f MB) in %.1fs" % (B, len(plan.off) - 1, len(plan.blob) / 1e6, time.time() - t0))
    words, build = LG.build_leg(ctx, plan, B, rows, log, calls=6 if rank == 0 else 1)   # C3 (BASELINE configs[2]) through the product build path
    if rank == 0:
        legs["build"] = build
        if not args.no_decode:
            legs["decode"] = LG.decode_leg(ctx, plan, words, B, log)
    or_reduce, or_state = None, None
    if args.or_union > 0:
        or_reduce, or_state = LG.or_reduce_leg(ctx, plan, B, args.fpr, args.or_union, world, log)
        legs["or_reduce"] = or_reduce
    if rank == 0 and world == 1 and args.ingest_blocks > 0:
        legs["ingest"] = LG.ingest_leg(ctx, min(args.ingest_blocks, B), rows, 0xB100F5EA4C4, workers, plan, words, args.fpr, log)

    exprs = synth.make_queries(NQ, args.workload, seed=1234)
    terms_per_query = 8 if args.workload == "c4" else 3
    cb = Q.compile_queries(exprs)
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=_lib.TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    bid = ctx.batch_create(terms, ops, poff)
    probe_kernel = "k_probe_terms_many" if np.bincount(np.asarray(kinds, dtype=np.int64), minlength=3).max() > 128 else "k_probe_terms"
    kstats = lambda tm: kernel_stats(tm, len(terms), probe_kernel)

    ft_bytes = int(sum((int(m) + 63) // 64 * 8 for m in plan.desc["m"][2::3]))
    R = args.replicas or max(2, int(np.ceil(2 * 256 * 2 ** 20 / max(ft_bytes, 1))))
    arenas = [ctx.arena_load(words, plan.desc) for _ in range(R)]
    log("arena: %d blocks, %.1f MB streamed per probe, %d replicas; batch: %d queries, %d distinct terms"
        % (B, ft_bytes / 1e6, R, NQ, len(terms)))

    # ---- correctness spot check against the oracle (outside the timed region) ----
    got = ctx.probe_batch(arenas[0], bid, NQ, B)
    if not args.no_check and rank == 0:
        from oracle import oracle as O
        # a RANDOM sample of the batch, checked by the oracle's tree-walking evaluator (query_exec.go:89-159 restated over the
        # expression trees: no postfix, none of the product's lowering in between)
        nchk = min(64, NQ)
        sel = np.sort(np.random.default_rng(20260927).choice(NQ, size=nchk, replace=False))
        want = O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), [exprs[int(i)] for i in sel])
        if not np.array_equal(got[sel], want):
            sys.exit("survivor sets differ from the oracle — refusing to report a number")
        # a grouped launch (several arenas behind one dispatch) must return exactly what one launch per arena returns
        many = ctx.probe_many(arenas[: min(R, 5)], bid, 0, NQ, [B] * min(R, 5))
        if not all(np.array_equal(m, got) for m in many):
            sys.exit("grouped probe differs from the single-arena probe")
        log("check: %d randomly chosen queries bit-exact vs the tree-walking oracle; grouped launch identical; %.2f%% of (query, block) pairs survive"
            % (nchk, 100.0 * sum(bin(int(x)).count("1") for x in got.ravel()) / (NQ * B)))

    # One step = one arena probed once.  Consecutive steps are handed to bsg_probe_many together (the library covers up to
    # --group arenas with one dispatch).  The K steps twice: bare (the headline: what a caller who asks for no timestamps pays), then
    # the same K steps with the dispatch's own start/stop timestamps on every launch (BSG_PROBE_TIMED: hipExtLaunchKernelGGL events on
    # the library's stream = what a rocprofv3 kernel trace reports) — the roofline's kernel durations come from these and from the
    # sampling pass below.  A profiled dispatch costs the host ~8 us more than a bare one.
    pr = Prober(ctx, bid, world, log, barrier)
    G0 = max(1, min(args.steps, args.group))          # arenas per dispatch in the timed region
    per_call = max(G0, (args.group // G0) * G0)
    make = lambda i: [arenas[i % R]]
    # untimed setup: one call of the timed region's shape sizes the library's verdict / survivor scratch
    pr.run(pr.plan([make(i) for i in range(min(per_call, args.steps))], per_call), 0)
    ctx.sync()
    if args.events_in_headline:
        elapsed, tm = pr.measure(make, args.steps, args.warmup, per_call, timed=True)
        clock = dict(pr.last_closing)
    else:
        elapsed, _ = pr.measure(make, args.steps, args.warmup, per_call, timed=False)
        clock = dict(pr.last_closing)
        _, tm = pr.measure(make, args.steps, min(args.warmup, 4), per_call, timed=True)
    timed_region = kstats(tm)

    # the same steps with the host-side gather inside the timed region: survivors of every step DMA-ed into a shared,
    # page-locked host segment (a ring: only the last calls' survivors are kept) ...
    wps = NQ * ((B + 63) // 64)
    ring_steps = min(max(args.steps, 1), 2 * per_call)
    row_words = wps * ring_steps
    sh = SharedHost(ctx, row_words * 8 + NQ * 4 * ring_steps, rank, world, "c2")
    h_elapsed, _ = pr.measure(make, args.steps, min(args.warmup, 4), per_call, timed=False, out=Ring(sh.mine[:row_words]), words_per_step=wps)
    if rank == 0 and not args.no_check and not np.array_equal(sh.part(0)[:wps].reshape(NQ, -1), got):
        sys.exit("survivors delivered to the shared host segment differ from the direct probe")
    page_locked = sh.registered
    # ... and as survivor ROWS (bsg_probe_many_rows): a 4-byte header per query + block ids / words only where the row needs them,
    # written by the device into the same page-locked segment — the north star's "host-side gather of surviving block IDs"
    r_elapsed = rows_tags = None
    if page_locked:
        from bloomsearch_amd.gpu import rows_to_dense
        hdr_ring = sh.mine[row_words:].view(np.uint32)[: NQ * ring_steps]
        r_elapsed, _ = pr.measure(make, args.steps, min(args.warmup, 4), per_call, timed=False, out=Ring(sh.mine[:row_words]), words_per_step=wps,
                                  hdr=Ring(hdr_ring), hdr_per_step=NQ)
        if rank == 0:
            h0 = sh.part(0)[row_words:].view(np.uint8)[:NQ] if C.ROWS_PACKED else sh.part(0)[row_words:].view(np.uint32)[:NQ]      # (packed: byte headers)
            if not args.no_check and not np.array_equal(rows_to_dense(h0, sh.part(0)[:wps], B, packed=C.ROWS_PACKED), got):
                sys.exit("survivor rows delivered to the shared host segment do not expand to the direct probe's bitsets")
            rows_tags = [int(x) for x in np.bincount(h0 >> (6 if C.ROWS_PACKED else 30), minlength=4)]
    sh.close()

    # ---- kernel sampling beyond the timed region: the driver's --steps may cover a single dispatch, the number of record
    # must not depend on it.  Same launch shape (G0 arenas per dispatch, rotating replicas), every dispatch timestamped.
    samples, all_t = {}, tm
    if args.samples > 0:
        _, t_s = pr.measure(make, args.samples * G0, G0, G0)      # n dispatches of each kernel, one call (one group) each
        samples, all_t = kstats(t_s), merge_timing(t_s, tm)
    allk = kstats(all_t)

    if not args.no_single and world == 1:
        ctx.set_probe_group(1)
        _, t1 = pr.measure(make, 64, 8, 64, nofuse=True)
        ctx.set_probe_group(args.group)
        single = kstats(t1)
        legs["roofline_single_launch"] = dict(single.get(probe_kernel, {}), bound="hbm", kernel=probe_kernel, peak=HBM_PEAK_GBPS, unit="GB/s",
                                              eval_kernel_ms=single.get("k_eval_programs", {}).get("kernel_ms"),
                                              note="one 1 000-block arena (35 MB) per dispatch: ~half of such a launch is dispatch ramp + completion")
    if rank == 0 and world == 1:
        if not args.no_q1:
            legs["q1"] = LG.q1_latency(ctx, arenas[0], B, None, log)
        if not args.no_concurrent:
            legs["concurrent_queries"] = LG.concurrent_queries_leg(ctx, arenas, B, exprs, got, log)
    if args.scaled > 1 and world == 1:
        for a in arenas[1:]:
            ctx.arena_free(a)
        arenas = arenas[:1]
        legs["roofline_scaled"] = dict(LG.scaled_leg(ctx, pr, args, plan, words, B, NQ, ft_bytes, len(terms), terms_per_query, log),
                                       bound="hbm", kernel=probe_kernel, peak=HBM_PEAK_GBPS, unit="GB/s",
                                       note="C2' of SURVEY 8d: same filters replicated x%d at distinct addresses, one arena, one launch" % args.scaled)
    for a in arenas:
        ctx.arena_free(a)
    arenas = []
    if rank == 0 and world == 1 and not args.no_big_filters:
        legs["big_filters"] = LG.big_filter_leg(ctx, args, log)
    c4 = None
    if args.c4_files > 0:
        c4 = legs["c4"] = LG.c4_leg(ctx, args, rank, world, workers, log, barrier, headline=world > 1)

    probes_per_step = NQ * B * terms_per_query * world
    h = None
    if rank == 0:
        # roofline of the dominant kernel of the timed region: algorithmic bytes per launch (streaming regime) =
        # every referenced bitset of the launch's arenas once + the term table; over the mean of the dispatch's own
        # start/stop timestamps, timed region and sampling pass together (same launch shape).
        dom = dominant_kernel(tm, probe_kernel)
        k = allk.get(dom) or {}
        traffic, traffic_src = LG.traffic_from_profiles(dom, k, args, B)
        copy_gbps = C.measured_copy_gbps(log)
        delivered = probes_per_step * args.steps / min(h_elapsed, r_elapsed or h_elapsed)
        h = {"metric": "block-bloom probes/sec", "value": probes_per_step * args.steps / elapsed, "unit": "probes/s", "n_gpus": world,
             "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
             "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
             "config": {"workload": "%s probe, survivors device-resident: %d rows/block x %d blocks per GPU, Q=%d %s, fpr %g; %d address-distinct "
                                    "arena replicas rotated per step, %d arenas (steps) per dispatch"
                                    % ("C4" if args.workload == "c4" else "C2", rows, B, NQ,
                                       "8-term Or(FieldToken)" if args.workload == "c4" else "3-term And(FieldToken)", args.fpr, R, G0),
                        "blocks_per_gpu": B, "queries": NQ, "distinct_terms": int(len(terms)), "probes_per_step": probes_per_step,
                        "sharding": "round-robin blocks, no collective"},
             "roofline": {"bound": "hbm", "kernel": dom, "achieved": k.get("achieved"), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": k.get("frac"), "traffic": traffic, "traffic_source": traffic_src,
                          "algorithmic_bytes_per_launch": k.get("algorithmic_bytes_per_launch"), "kernel_ms": k.get("kernel_ms"),
                          "samples": k.get("samples"), "arenas_per_launch": k.get("arenas_per_launch"), "copy_gbps": copy_gbps,
                          "frac_of_copy": (k.get("achieved") / copy_gbps) if (copy_gbps and k.get("achieved")) else None},
             "clock": dict(clock, barrier="host flags in shared memory after each rank's device synchronize; closing barrier INSIDE value/ms_per_step"),
             "value_survivors_delivered_to_host": delivered,
             "survivors_note": "value: survivor bitsets stay in HBM; value_survivors_delivered_to_host: the same K steps with every query's "
                               "survivor rows (bsg_probe_many_rows) written into page-locked host memory, PCIe-inclusive"}
        legs["c2_probe"] = {"timed_region": timed_region, "sampling_pass": samples, "all": allk, "clock_note": C.CLOCK_NOTE,
                            "valu_issue": LG.valu_issue_from_profiles(dom, k, B) if dom == "k_probe_terms_many" else None,
                            "host_gather": {"ms_per_step": h_elapsed / args.steps * 1e3, "value": probes_per_step * args.steps / h_elapsed,
                                            "survivor_bytes_per_step_per_gpu": wps * 8, "page_locked": page_locked,
                                            "rows": None if r_elapsed is None else {
                                                "api": "bsg_probe_many_rows" + (" (BSG_PROBE_ROWS_PACKED)" if C.ROWS_PACKED else ""), "ms_per_step": r_elapsed / args.steps * 1e3,
                                                "value": probes_per_step * args.steps / r_elapsed, "rows_by_tag_none_all_list_dense": rows_tags}}}
        if world > 1 and c4:
            # N > 1: the headline is BASELINE configs[3] — C4, STRONG scaling (10 000 blocks in total, block b on rank b % N, the 8-term
            # Or batch, exactly --steps timed steps); the weak-scaling C2 figures above move to the legs.
            ck = (c4.get("kernels") or {}).get(c4.get("dominant_kernel")) or {}
            legs["c2_weak"] = {key: h[key] for key in ("value", "value_survivors_delivered_to_host", "steps", "warmup", "ms_per_step", "clock",
                                                       "scaling", "config", "roofline")}
            h.update({"value": c4["value"], "value_survivors_delivered_to_host": (c4["host_gather"].get("rows") or c4["host_gather"])["value"],
                      "steps": c4["steps"], "warmup": c4["warmup"], "ms_per_step": c4["ms_per_step"], "scaling": "strong",
                      "clock": dict(c4["clock"], barrier=h["clock"]["barrier"]),
                      "config": {"workload": c4["workload"] + "; survivors device-resident", "blocks_total": c4["blocks_total"], "queries": NQ,
                                 "probes_per_step": c4["probes_per_step"], "sharding": "block b -> rank b % N, no collective",
                                 "curve": "strong scaling of BASELINE configs[3]: compare with c4.value of the N = 1 line"},
                      "roofline": {"bound": "hbm", "kernel": c4.get("dominant_kernel"), "achieved": ck.get("achieved"), "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": ck.get("frac"), "traffic": None,
                                   "algorithmic_bytes_per_launch": ck.get("algorithmic_bytes_per_launch"), "kernel_ms": ck.get("kernel_ms"),
                                   "samples": ck.get("samples"), "arenas_per_launch": ck.get("arenas_per_launch"), "copy_gbps": copy_gbps,
                                   "note": "rank 0's dispatches of the timed region; every rank's kernel time: bench_legs.json c4.per_rank"}})
            h["clock"].pop("note", None)
        if args.cpu_budget > 0 and world == 1:
            base, cpu_out, nq = cpu_baseline(words, plan.desc, cb, ops, poff, B, args.cpu_budget, log, terms_per_query)
            if not args.no_check and not np.array_equal(cpu_out, got[:nq]):
                sys.exit("CPU baseline survivors differ from the GPU's")
            h["cpu_baseline"] = go_reference_baseline(rows, min(B, 100), log) or base
            if h["cpu_baseline"] is not base:
                legs["cpu_baseline_port"] = base

    import threading
    emitted = threading.Lock()

    def emit():
        if rank == 0 and emitted.acquire(blocking=False):
            write_legs(legs, log)
            print(headline(h, legs), file=json_out, flush=True)
    closing_legs(emit, legs, ctx, args, rank, world, local_rank, plan, words, block_ids, rows, terms, ops, poff, got, or_reduce, or_state, log)
    C.stop_pool()
    barrier.close()
    ctx.batch_free(bid)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
