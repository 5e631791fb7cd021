"""What a bsg_query caller spends before it reaches the combiner (GPU box): python tools/conc_dry.py
lab key 12 = 2: the call returns after validation, hashing, arena lookup and the compact layout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bloomsearch_amd import conc, query as Q, synth
from bloomsearch_amd.arena import plan_blocks
from bloomsearch_amd.gpu import Context

blocks = [synth.block_entry_sets(b * 100, 100) for b in range(1000)]
plan = plan_blocks(blocks, 0.001)
with Context((0,)) as ctx:
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    arenas = [ctx.arena_load(words, plan.desc) for _ in range(12)]
    exprs = synth.make_queries(256, "c2", seed=1234)
    expected = np.zeros((256, 16), dtype=np.uint64)
    ctx.set_lab(12, 2)
    for pool in (arenas[:1], arenas):
        for T in (1, 16, 64, 256):
            r = conc.run(ctx, exprs, pool, 1000, expected, T, 0.3, 1)
            print("dry T=%3d arenas %2d: %.3g calls/s, cpu %.2f us/call (%.1f busy), p50 %.1f us" % (T, len(pool), r["queries_per_s"], r["cpu_us_per_call"], r["cpus_busy"], r["p50_us"]), flush=True)
