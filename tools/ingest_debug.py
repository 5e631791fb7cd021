import sys
sys.path.insert(0, '.')
import numpy as np
from bloomsearch_amd import ingest as I, synth
from bloomsearch_amd.gpu import Context
ctx = Context((0,))
for n in (1, 2, 64, 65, 256, 257, 700):
    rows = synth.rows_json(0, n)
    res = I.device_ingest(ctx, [rows], 0.001)
    print(n, res.counts.tolist(), len(res.fallback_rows))
rows = [b'{"level":"info"}'] * 700
res = I.device_ingest(ctx, [rows], 0.001)
print('same row x700', res.counts.tolist())
rows = [b'{"level":"info"}'] * 64
res = I.device_ingest(ctx, [rows], 0.001)
print('same row x64', res.counts.tolist())
