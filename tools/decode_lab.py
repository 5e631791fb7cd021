import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np
from benchlib import common as bench
from bloomsearch_amd.arena import plan_blocks
from bloomsearch_amd.gpu import Context
ctx = Context((0,))
B = 1000
blocks = bench.generate_blocks(np.arange(B, dtype=np.int64), 10000, 0xB100F5EA4C4, 32)
plan = plan_blocks(blocks, 0.001)
secs = ctx.build_sections(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
nb = sum(len(x) for x in secs)
for pieces in (1, 4):
    ctx.set_lab(4, pieces)
    ms = []
    for _ in range(6):
        sid, st = ctx.arena_load_sections(secs)
        assert not st.any()
        ms.append(ctx.last_kernel_ms()[2])
        ctx.arena_free(sid)
    print("pieces %d: decode kernels %.1f us (min %.1f) for %.1f MB = %.0f GB/s (read + written)" % (pieces, np.median(ms) * 1e3, min(ms) * 1e3, nb / 1e6, 2 * nb / np.median(ms) / 1e6))
