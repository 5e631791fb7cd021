"""One bsg_build call over N synthetic blocks (for profiling k_build in isolation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bloomsearch_amd import synth
from bloomsearch_amd.arena import plan_blocks
from bloomsearch_amd.gpu import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
blocks = [synth.block_entry_sets(b * 10000, 10000) for b in range(n)]
plan = plan_blocks(blocks, 0.001)
with Context((0,)) as ctx:
    for _ in range(3):
        ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        print("k_build %.1f us for %d entries" % (ctx.last_kernel_ms()[0] * 1e3, len(plan.off) - 1))
