"""Fixed per-flush cost of the device ingest path: 1 000-row flushes timed call by call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bloomsearch_amd import host as Hst, ingest as I, synth
from bloomsearch_amd.gpu import Context

ctx = Context((0,))
rows = synth.rows_json(0, 50000)
for mode in (False, True):
    for buffered in (1000, 10000):
        e = Hst.Engine(ctx, MaxBufferedRows=buffered, MaxBufferedBytes=1 << 40, DeviceIngest=mode)
        t0 = time.perf_counter()
        for i in range(0, len(rows), 1000):
            e.ingest_rows(rows[i:i + 1000])
        e.flush()
        dt = time.perf_counter() - t0
        print("DeviceIngest=%s MaxBufferedRows=%d: %.2f us/row, %d files" % (mode, buffered, dt / len(rows) * 1e6, len(e.describe()["files"])))
        e.close()
acc = np.zeros(5)
n = 30
for it in range(n):
    r = rows[it * 1000:(it + 1) * 1000]
    t = [time.perf_counter()]
    ing = ctx.ingest_rows(r, [0, 1000], [0], 1, flags=1)
    t.append(time.perf_counter())
    ctx.ingest_fallback_rows(ing)
    t.append(time.perf_counter())
    counts, status = ctx.ingest_finish(ing, 2)
    t.append(time.perf_counter())
    desc, n_words = I.plan_desc(counts, 0.001)
    secs, a, b = ctx.ingest_build_sections(ing, desc, arenas=True)
    t.append(time.perf_counter())
    ctx.ingest_free(ing)
    ctx.arena_free(a)
    ctx.arena_free(b)
    t.append(time.perf_counter())
    if it >= 5:
        acc += np.diff(t)
print("per 1000-row flush (us): ingest_rows %.0f, fallback_rows %.0f, finish %.0f, build_sections+arenas %.0f, free %.0f"
      % tuple(acc / (n - 5) * 1e6))
