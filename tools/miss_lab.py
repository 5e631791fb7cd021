"""What a miss of the file-arena cache pays: bsg_arena_load_sections + bsg_arena_free of 1 / 8 / 64 small sections, one caller.
Round 6: 3.2 ms whatever the size (a hipStreamCreate / Destroy pair per load) -> 0.10 ms with the copy stream and events reused."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd.gpu import Context
from tests.test_arena_cache_gpu import make_file
rng = np.random.default_rng(1)
secs, *_ = make_file(rng, 64, max_tokens=1500)
with Context((0,)) as ctx:
    for n in (1, 8, 64):
        s = secs[:n]
        for _ in range(5):
            a, _ = ctx.arena_load_sections(s); ctx.arena_free(a)
        t0 = time.perf_counter(); ids = []
        for _ in range(200):
            ids.append(ctx.arena_load_sections(s)[0])
        t1 = time.perf_counter()
        for a in ids: ctx.arena_free(a)
        t2 = time.perf_counter()
        print("%2d sections (%d KB): load %.1f us, free %.1f us" % (n, sum(map(len, s)) // 1024, (t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
