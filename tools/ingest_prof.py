"""Standalone device-ingest run for profiling: N blocks of synthetic JSON rows through bsg_ingest_*."""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from benchlib import common as bench
if os.environ.get('BSG_LAB_LIB'):
    from bloomsearch_amd import _lib
    _lib.LIB_PATH = os.environ['BSG_LAB_LIB']
from bloomsearch_amd import ingest as I
from bloomsearch_amd.gpu import Context

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n_par = int(sys.argv[5]) if len(sys.argv) > 5 else 1
# in-process generation: a forked pool under rocprofv3 hangs at exit (the children inherit the tool)
parts = [bench._gen_rows((b, rows, 0xB100F5EA4C4)) for b in range(n_blocks)]
blob = np.frombuffer(b"".join(p[0] for p in parts), dtype=np.uint8)
lens = np.concatenate([p[1] for p in parts])
off = np.zeros(len(lens) + 1, dtype=np.uint64)
np.cumsum(lens, out=off[1:])
first = np.arange(n_blocks + 1, dtype=np.uint32) * rows
ctx = Context((0,))
for rep in range(reps):
    t0 = time.time()
    ing = ctx.ingest_rows((blob, off), first, np.zeros(n_blocks, dtype=np.uint32) if n_par else None, n_par, flags=flags)
    t1 = time.time()
    counts, status = ctx.ingest_finish(ing, n_blocks + n_par)
    t2 = time.time()
    desc, n_words = I.plan_desc(counts, 0.001)
    t3 = time.time()
    ctx.ingest_build(ing, desc, n_words)
    t4 = time.time()
    st = ctx.ingest_stats(ing)
    ctx.ingest_free(ing)
    t5 = time.time()
    from bloomsearch_amd import query as Q
    hits, handed_back = ctx.match_rows((blob, off), Q.CompiledMatcher(Q.FieldToken("level", "error")))
    print("    match: k_match_rows %.2f ms, %d matches, %d handed back, %.1f ms wall" % (ctx.last_match_ms(), int(hits.sum()), len(handed_back), (time.time() - t5) * 1e3))
    print("    host wall: ingest_rows %.1f ms (kernel %.1f), finish %.1f ms (kernel %.1f), plan_desc %.1f ms, build %.1f ms (kernel %.1f), free %.1f ms"
          % ((t1 - t0) * 1e3, st.ms_walk, (t2 - t1) * 1e3, st.ms_union, (t3 - t2) * 1e3, (t4 - t3) * 1e3, st.ms_build, (t5 - t4) * 1e3))
    print("rep %d: %d rows %.0f MB  walk %.2f ms  union %.2f ms  build %.2f ms  grows %d  fallback %d  tables %.0f MB  e2e %.3fs  file counts %s"
          % (rep, n_blocks * rows, st.row_bytes / 1e6, st.ms_walk, st.ms_union, st.ms_build, st.table_grows, st.n_fallback_rows,
             st.table_bytes / 1e6, time.time() - t0, counts[-1].tolist()))
