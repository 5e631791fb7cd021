"""Combined bsg_query callers beside a flush worker that keeps the device busy (bsg_ingest_rows of ~1 GB over and over): rate, latency and
PROCESSOR time with the collector's doorbell wait at its default (poll <= 50 us, then sleep on a blocking-sync event) and at round 5's
20 ms spin (bsg_set_lab key 25 = 20000).  python tools/conc_busy.py"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd import conc, query as Q, synth          # noqa: E402
from bloomsearch_amd.arena import plan_blocks                 # noqa: E402
from bloomsearch_amd.gpu import Context                       # noqa: E402

with Context((0,)) as ctx:
    plan = plan_blocks([synth.block_entry_sets(b * 400, 400) for b in range(200)], 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aids = [ctx.arena_load(words, plan.desc) for _ in range(6)]
    exprs = synth.make_queries(48, "c2", seed=78)
    ctx.set_lab(12, 0)
    expected = np.stack([ctx.query([aids[0]], [200], Q.compile_queries([e]))[0][0] for e in exprs])
    ctx.set_lab(12, 1)
    base = synth.rows_json(0, 4000)
    reps = 1000
    blob = np.tile(np.frombuffer(b"".join(base), dtype=np.uint8), reps)
    lens = np.tile(np.asarray([len(r) for r in base], dtype=np.uint64), reps)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    first = np.asarray([0, len(lens)], dtype=np.uint32)
    for spin_us, name in ((0, "adaptive poll (4 x the mean polled wait, 50 us .. 1 ms), then sleep on an event (default)"), (50, "poll 50 us, then sleep"), (20000, "poll 20 ms (round 5)")):
        ctx.set_lab(25, spin_us)
        quiet = conc.run(ctx, exprs, aids, 200, expected, n_threads=64, seconds=0.5)
        stop, walks = threading.Event(), []

        def flush_worker():
            while not stop.is_set():
                ing = ctx.ingest_rows((blob, off), first, np.zeros(1, dtype=np.uint32), 1, flags=1)
                walks.append(ctx.ingest_stats(ing).ms_walk)
                ctx.ingest_free(ing)
        t = threading.Thread(target=flush_worker)
        t.start()
        while not walks:
            threading.Event().wait(0.05)
        busy = conc.run(ctx, exprs, aids, 200, expected, n_threads=64, seconds=2.0)
        stop.set()
        t.join()
        assert busy["errors"] == 0 and busy["mismatches"] == 0
        print("%s\n   quiet device: %.3g q/s, p50 %.0f us, p99 %.0f us, %.1f us of processor time per call, %.1f CPUs busy"
              % (name, quiet["queries_per_s"], quiet["p50_us"], quiet["p99_us"], quiet["cpu_us_per_call"], quiet["cpus_busy"]))
        print("   beside bsg_ingest_rows (%d calls, k_ingest_rows %.1f ms each): %.3g q/s, p50 %.0f us, p99 %.0f us, %.1f us of processor time per call, %.1f CPUs busy"
              % (len(walks), float(np.median(walks)), busy["queries_per_s"], busy["p50_us"], busy["p99_us"], busy["cpu_us_per_call"], busy["cpus_busy"]))
