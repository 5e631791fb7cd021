"""Where the end-to-end time of a 10 M-row device ingest goes (GPU box): BSG_LAB_TRACE=1 python tools/ingest_e2e.py [blocks]"""
import os, sys, time
import multiprocessing as mp
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd import synth, ingest as I
from bloomsearch_amd.gpu import Context


def gen(b):
    rs = synth.rows_json(b * 10000, 10000)
    return b"".join(rs), np.asarray([len(r) for r in rs], dtype=np.uint32)


nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with mp.get_context("fork").Pool(min(64, os.cpu_count())) as pool:
    parts = pool.map(gen, range(nb), chunksize=4)
blob = np.frombuffer(b"".join(p[0] for p in parts), dtype=np.uint8)
lens = np.concatenate([p[1] for p in parts])
off = np.zeros(len(lens) + 1, dtype=np.uint64)
np.cumsum(lens, out=off[1:])
first = np.arange(nb + 1, dtype=np.uint32) * 10000
ctx = Context((0,))
pinned = ctx.pinned_array(len(blob))
pinned[:] = blob
for rep in range(3):
    for name, src in (("pageable", blob), ("pinned", pinned)):
        t0 = time.perf_counter()
        ing = ctx.ingest_rows((src, off), first, np.zeros(nb, dtype=np.uint32), 1, flags=1)
        t1 = time.perf_counter()
        counts, status = ctx.ingest_finish(ing, nb + 1)
        t2 = time.perf_counter()
        desc, n_words = I.plan_desc(counts, 0.001)
        t3 = time.perf_counter()
        words = ctx.ingest_build(ing, desc, n_words)
        t4 = time.perf_counter()
        st = ctx.ingest_stats(ing)
        ctx.ingest_free(ing)
        print("%s rep %d: ingest_rows %.1f ms (walk kernels %.1f) finish %.1f ms (union %.1f) plan %.1f ms build %.1f ms (kernel %.1f) total %.1f ms"
              % (name, rep, (t1 - t0) * 1e3, st.ms_walk, (t2 - t1) * 1e3, st.ms_union, (t3 - t2) * 1e3, (t4 - t3) * 1e3, st.ms_build, (t4 - t0) * 1e3), flush=True)
