"""Lab: the folded probe + evaluation dispatch (k_probe_eval) against k_probe_terms + k_eval_programs on the C2 shape.
python tools/fold_lab.py [arenas_per_call] [workload]  — prints kernel time and wall time per call for several helper counts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bloomsearch_amd import _lib, query as Q, synth
from bloomsearch_amd.arena import plan_blocks
from bloomsearch_amd.gpu import Context
from benchlib import common as bench


def main():
    per_call = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
    ctx = Context((0,))
    B, rows, NQ = 1000, 10000, 4096
    blocks = bench.generate_blocks(np.arange(B, dtype=np.int64), rows, 0xB100F5EA4C4, 32)
    plan = plan_blocks(blocks, 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    cb = Q.compile_queries(synth.make_queries(NQ, workload, seed=1234))
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=_lib.TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    bid = ctx.batch_create(terms, ops, poff)
    R = 16
    arenas = [ctx.arena_load(words, plan.desc) for _ in range(R)]
    ref = ctx.probe_batch(arenas[0], bid, NQ, B, flags=_lib.PROBE_NOFUSE)
    ids = np.ascontiguousarray([arenas[i % R] for i in range(per_call)], dtype=np.uint64)
    settings = [("two kernels", 0), ("fold K=1", 1), ("fold K=2", 2), ("fold K=4", 4), ("fold K=8", 8), ("fold K=16", 16)]
    if os.environ.get("BSG_LAB_FOLD_SKIP"):
        settings = [("two kernels", 0), ("fold, evaluation skipped", 8)]
    for name, k in settings:
        ctx.set_lab(11, k)
        if not os.environ.get("BSG_LAB_FOLD_SKIP") and not os.environ.get("BSG_LAB_FOLD"):
            got = ctx.probe_batch(arenas[1], bid, NQ, B)
            assert np.array_equal(got, ref), name
        for timed in (True, False):
            flags = (_lib.PROBE_TIMED if timed else 0) | _lib.PROBE_ASYNC
            for _ in range(5):
                ctx.probe_many(ids, bid, flags)
            ctx.sync()
            ctx.timing_read(reset=True)
            n = 30
            walls = []
            for _ in range(n):
                t0 = time.perf_counter()
                ctx.probe_many(ids, bid, flags)
                ctx.sync()
                walls.append(time.perf_counter() - t0)
            tm = ctx.timing_read()
            if timed:
                kern = (tm.ms_terms_kernel + tm.ms_eval_kernel + tm.ms_folded_kernel + tm.ms_fused_kernel) / n * 1e3
                detail = "probe %.1f + eval %.1f" % (tm.ms_terms_kernel / n * 1e3, tm.ms_eval_kernel / n * 1e3) if tm.n_probes else "folded %.1f" % (tm.ms_folded_kernel / n * 1e3)
                print("%-28s kernels %.1f us per call (%s); wall %.1f us per call (timed launches)" % (name, kern, detail, np.median(walls) * 1e6), flush=True)
            else:
                print("%-28s wall %.1f us per call = %.2f us per arena (untimed launches)" % (name, np.median(walls) * 1e6, np.median(walls) * 1e6 / per_call), flush=True)


if __name__ == "__main__":
    main()
