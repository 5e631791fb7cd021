"""The resident file-arena cache under T native callers (tools/native/conc_driver.cpp::cache_run): hits, misses, widenings, evictions and
call rate for budgets that hold all / a quarter of the files, with and without tombstones while leased.  python tools/cache_lab.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd import conc, query as Q          # noqa: E402
from bloomsearch_amd.gpu import Context               # noqa: E402
from tests.test_arena_cache_gpu import expected_survivors, load_and_publish, make_file          # noqa: E402

rng = np.random.default_rng(2026)
n_files, nb = 32, 64
made = [make_file(rng, nb, max_tokens=1500) for _ in range(n_files)]
vocab = made[0][4]
exprs = [Q.And(Q.Token(vocab[int(rng.integers(0, 400))]), Q.Token(vocab[int(rng.integers(0, 5000))])) for _ in range(16)] + [Q.Token(vocab[int(i)]) for i in rng.integers(0, 5000, size=16)]
files = [m[0] for m in made]
expected = [expected_survivors(m[1], m[2], exprs) for m in made]
with Context((0,)) as ctx:
    lease, _, _ = load_and_publish(ctx, b"probe", files[0], range(nb))
    one = ctx.arena_cache_stats()["resident_bytes"]
    ctx.file_arena_release(lease)
    ctx.file_arena_forget(b"probe")
    print("%d files x %d blocks, one whole-file arena = %.0f KB on the device; every candidate verdict of every call checked against the tree oracle" % (n_files, nb, one / 1e3))
    for T in (16, 64, 256):
        for frac, name in ((2.0, "holds every file"), (0.25, "holds a quarter")):
            for forget in (0, 29):
                ctx.set_arena_budget(int(one * n_files * frac))
                ctx.arena_cache_stats(reset=True)
                r = conc.cache_run(ctx, exprs, files, expected, n_threads=T, seconds=1.0, forget_every=forget, seed=T)
                st = ctx.arena_cache_stats()
                assert r["errors"] == 0 and r["mismatches"] == 0 and st["leases"] == 0 and st["leased_dead_bytes"] == 0, (r, st)
                print("T=%3d budget %-17s forget every %2d: %7.0f calls/s, hit rate %.3f, published %5d, widenings %5d, evictions %5d, forgotten %4d, rejected dirty/over/narrower %d/%d/%d, resident %.1f MB of %.1f"
                      % (T, name, forget, r["calls"] / r["seconds"], r["hits"] / max(r["hits"] + r["misses"], 1), st["published"], st["widenings"], st["evictions"],
                         st["forgotten"], st["rejected_dirty"], st["rejected_over_budget"], st["rejected_narrower"], st["resident_bytes"] / 1e6, st["budget_bytes"] / 1e6))
                for f in range(n_files):
                    ctx.file_arena_forget(bytes([f, 0, 0, 0]))
