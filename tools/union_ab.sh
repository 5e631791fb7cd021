#!/bin/bash
# A/B of the file-level union at NB (default 300) blocks x 10 000 rows: LDS partitions (default) vs global hash tables (lab key 9 = 1),
# and the default child-table capacity (4 slots per row, load ~0.3) vs a hint of 32 768 slots (load ~0.6)
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from benchlib import common as bench
from bloomsearch_amd import ingest as I
from bloomsearch_amd.gpu import Context
n_blocks, rows = int(os.environ.get("NB", "300")), 10000
parts = [bench._gen_rows((b, rows, 0xB100F5EA4C4)) for b in range(n_blocks)]
blob = np.frombuffer(b"".join(p[0] for p in parts), dtype=np.uint8)
lens = np.concatenate([p[1] for p in parts])
off = np.zeros(len(lens) + 1, dtype=np.uint64); np.cumsum(lens, out=off[1:])
first = np.arange(n_blocks + 1, dtype=np.uint32) * rows
ctx = Context((0,))
hint_small = np.zeros(n_blocks * 3, dtype=np.uint32); hint_small[1::3] = 32768; hint_small[2::3] = 32768      # child tables at load ~0.6 instead of ~0.3
for mode, hint in ((0, None), (1, None), (0, hint_small), (0, None), (1, None), (0, hint_small)):
    ctx.set_lab(9, mode)
    ing = ctx.ingest_rows((blob, off), first, np.zeros(n_blocks, dtype=np.uint32), 1, slots_hint=hint, flags=1)
    counts, status = ctx.ingest_finish(ing, n_blocks + 1)
    desc, n_words = I.plan_desc(counts, 0.001)
    w = ctx.ingest_build(ing, desc, n_words)
    st = ctx.ingest_stats(ing)
    ctx.ingest_free(ing)
    print("hint %s mode %d (%s): walk %.2f ms union %.3f ms build %.2f ms tables %.0f MB grows %d file counts %s xor %016x" % ("32768" if hint is not None else "default", mode, "global tables" if mode else "LDS partitions", st.ms_walk, st.ms_union, st.ms_build, st.table_bytes / 1e6, st.table_grows, counts[-1].tolist(), int(np.bitwise_xor.reduce(w))))
PY
