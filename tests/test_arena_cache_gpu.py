"""The resident file-arena cache of the library (csrc/cache_api.inc; SURVEY 8 f2, reference analogue: blockFilterCursor re-reading
a file's filter region per query, file_format.go:511-662 + query_exec.go:546-615; tombstones: merge.go:178-185).

Policy under test: least recently used by BYTES against a budget; an arena somebody holds a lease on is never evicted; only clean
decodes become resident; a miss widens to the union of the block sets; a narrower publish never replaces a wider resident arena;
forget takes the file out at once and its last user frees it; nothing leaks.  Survivors are checked against the oracle's
tree-walking evaluator (query_exec.go:89-159 restated) throughout — a cache that hands out the wrong arena or the wrong rows
fails here, not just one that mis-counts."""
import numpy as np
import pytest

from bloomsearch_amd import _lib, query as Q
from bloomsearch_amd.gpu import Context
from oracle import oracle as O
from tests.helpers import device_free_bytes, make_random_arena, oracle_words, random_expression

pytestmark = pytest.mark.gpu


def key_of(b):
    return b * 4096 + 17          # the block's RowDataOffset stand-in (tools/native/conc_driver.cpp uses the same)


def make_file(rng, n_blocks, max_tokens=1500):
    """-> (sections: list[bytes] per block, oracle filters per block, block strings, vocab)"""
    plan, blocks_str, vocab = make_random_arena(rng, n_blocks, fpr=0.01, max_tokens=max_tokens, absent_frac=0.0)
    words = oracle_words(plan)
    desc = plan.desc.view(O.DESC_DTYPE)
    secs = []
    for b in range(n_blocks):
        fl = []
        for c in range(3):
            d = desc[b * 3 + c]
            nw = (int(d["m"]) + 63) // 64
            fl.append(O.Filter(int(d["m"]), int(d["k"]), words[int(d["word_off"]): int(d["word_off"]) + nw]))
        secs.append(O.encode_filter_section(fl))
    return secs, words, plan.desc, blocks_str, vocab


def expected_survivors(words, desc, exprs):
    return O.survivors_tree(words, desc.view(O.DESC_DTYPE), exprs)


def load_and_publish(ctx, key, secs, blocks, status_override=None):
    """Loads the sections of `blocks` (ascending block indices) and publishes them: (lease, arena, resident)."""
    aid, status = ctx.arena_load_sections([secs[b] for b in blocks])
    off = np.zeros(len(secs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in secs])
    lease, resident = ctx.file_arena_publish(key, aid, [key_of(b) for b in blocks], off[list(blocks)], off[[b + 1 for b in blocks]],
                                             status if status_override is None else status_override)
    return lease, aid, resident


def probe_rows(ctx, arena, n_arena_blocks, exprs, rows):
    """Survivor bit of every (query, candidate) through the leased arena: [len(exprs), len(rows)] bool."""
    cb = Q.compile_queries(exprs)
    got = ctx.query([arena], [n_arena_blocks], cb)[0]
    rows = np.asarray(rows, dtype=np.int64)
    return ((got[:, rows >> 6] >> (rows & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)


def want_bits(want, blocks):
    blocks = np.asarray(blocks, dtype=np.int64)
    return ((want[:, blocks >> 6] >> (blocks & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)


def test_policy_hit_widen_narrower_dirty_lru_forget_and_no_leak():
    rng = np.random.default_rng(606)
    nb = 40
    files = {k: make_file(rng, nb) for k in (b"f1", b"f2", b"f3", b"f4")}
    secs1, words1, desc1, strs1, vocab = files[b"f1"]
    exprs = [random_expression(rng, vocab, None) for _ in range(24)] + [Q.Token(strs1[3][1][0]), None]
    want1 = expected_survivors(words1, desc1, exprs)
    with Context((0,)) as ctx:
        ctx.arena_free(ctx.arena_load_sections(secs1)[0])            # (warms the library's scratch pool: what it keeps is not a leak)
        ctx.sync()
        free0 = device_free_bytes()
        ctx.set_arena_budget(1 << 40)
        # miss -> load a run of blocks -> publish -> resident
        run = list(range(5, 20))
        assert ctx.file_arena_acquire(b"f1", [key_of(b) for b in run]) == (0, 0, None)
        lease, arena, resident = load_and_publish(ctx, b"f1", secs1, run)
        assert resident and lease
        with pytest.raises(_lib.BloomGpuError, match="belongs to the file-arena cache"):
            ctx.arena_free(arena)                                    # the cache owns it now
        assert np.array_equal(probe_rows(ctx, arena, len(run), exprs, range(len(run))), want_bits(want1, run))
        ctx.file_arena_release(lease)
        with pytest.raises(_lib.BloomGpuError) as e:
            ctx.file_arena_release(lease)                            # double release
        assert e.value.code == _lib.BSG_E_NOTFOUND
        # hit on a subset: rows point into the resident arena
        sub = [6, 9, 19]
        l2, a2, rows = ctx.file_arena_acquire(b"f1", [key_of(b) for b in sub])
        assert l2 and a2 == arena and list(rows) == [1, 4, 14]
        assert np.array_equal(probe_rows(ctx, a2, len(run), exprs, rows), want_bits(want1, sub))
        ctx.file_arena_release(l2)
        # candidates the resident arena does not cover: miss; `have` names what is there; the caller loads the union (widening)
        other = [0, 1, 30]
        assert ctx.file_arena_acquire(b"f1", [key_of(b) for b in other])[0] == 0
        keys, sb, se = ctx.file_arena_have(b"f1")
        assert list(keys) == [key_of(b) for b in run] and all(int(e - s) == len(secs1[b]) for s, e, b in zip(sb, se, run))
        union = sorted(set(run) | set(other))
        l3, a3, res3 = load_and_publish(ctx, b"f1", secs1, union)
        assert res3 and a3 != arena
        st = ctx.arena_cache_stats()
        assert st["widenings"] == 1 and st["resident_files"] == 1 and st["hits"] == 1 and st["misses"] == 2
        with pytest.raises(_lib.BloomGpuError):                      # the replaced arena had no users: it is gone
            ctx.query([arena], [len(run)], Q.compile_queries(exprs[:1]))
        assert np.array_equal(probe_rows(ctx, a3, len(union), exprs, range(len(union))), want_bits(want1, union))
        # a NARROWER arena published meanwhile (a concurrent query that raced the widening) serves its own lease only
        l4, a4, res4 = load_and_publish(ctx, b"f1", secs1, [2, 3])
        assert not res4
        assert np.array_equal(probe_rows(ctx, a4, 2, exprs, [0, 1]), want_bits(want1, [2, 3]))
        ctx.file_arena_release(l4)
        with pytest.raises(_lib.BloomGpuError):
            ctx.query([a4], [2], Q.compile_queries(exprs[:1]))       # freed by its release
        l_f1, a_f1, _ = ctx.file_arena_acquire(b"f1", [key_of(b) for b in union])     # the wide one stayed; this lease stays open below
        assert a_f1 == a3
        ctx.file_arena_release(l3)
        st = ctx.arena_cache_stats()
        assert st["rejected_narrower"] == 1 and st["leases"] == 1
        # a DIRTY decode never becomes resident: a section with a flipped byte fails its CRC (ErrInvalidHash), the block gets nil filters
        secs2 = list(files[b"f2"][0])
        bad = bytearray(secs2[7]); bad[len(bad) // 2] ^= 0x40; secs2[7] = bytes(bad)
        l5, a5, res5 = load_and_publish(ctx, b"f2", secs2, range(nb))
        assert not res5 and ctx.arena_cache_stats()["rejected_dirty"] == 1
        ctx.file_arena_release(l5)
        assert ctx.file_arena_acquire(b"f2", [key_of(0)])[0] == 0    # next query re-reads (and, with clean bytes, recovers)
        l5, a5, res5 = load_and_publish(ctx, b"f2", files[b"f2"][0], range(nb))
        assert res5
        ctx.file_arena_release(l5)
        # LRU by bytes: a budget that holds f1 (wide) + f2 only; f3 arrives -> the least recently used file nobody uses leaves
        st = ctx.arena_cache_stats()
        two = st["resident_bytes"]
        ctx.set_arena_budget(two + 1000)
        lf2, af2, _ = ctx.file_arena_acquire(b"f2", [key_of(0)])     # f2 is now the most recently used, f1 the least ... but f1 is IN USE (l_f1)
        l6, a6, res6 = load_and_publish(ctx, b"f3", files[b"f3"][0], range(nb))
        assert res6
        st = ctx.arena_cache_stats()
        assert st["evictions"] == 0 and st["resident_files"] == 3 and st["resident_bytes"] > st["budget_bytes"]    # both older entries are in use: the budget waits
        ctx.file_arena_release(lf2)
        # the release that ends f2's last lease runs the eviction the budget was waiting for: f2 is unused and older than f3
        st = ctx.arena_cache_stats()
        assert st["evictions"] == 1 and st["resident_files"] == 2
        assert ctx.file_arena_acquire(b"f2", [key_of(0)])[0] == 0 and ctx.file_arena_have(b"f2")[0].size == 0
        assert ctx.file_arena_have(b"f1")[0].size == len(union)
        ctx.file_arena_release(l6)
        # an arena beyond the whole budget serves its query and is not kept
        ctx.set_arena_budget(1000)
        assert ctx.arena_cache_stats()["resident_files"] == 1        # f3 evicted; f1 still leased
        l7, a7, res7 = load_and_publish(ctx, b"f4", files[b"f4"][0], range(nb))
        assert not res7 and ctx.arena_cache_stats()["rejected_over_budget"] == 1
        ctx.file_arena_release(l7)
        # FORGET while in use (the file was merged away): out of the table at once, the lease keeps probing, the release frees
        ctx.set_arena_budget(1 << 40)
        ctx.file_arena_forget(b"f1")
        assert ctx.file_arena_acquire(b"f1", [key_of(5)])[0] == 0 and ctx.arena_cache_stats()["resident_files"] == 0
        assert np.array_equal(probe_rows(ctx, a3, len(union), exprs, range(len(union))), want_bits(want1, union))
        st = ctx.arena_cache_stats()
        assert st["leases"] == 1 and st["leased_dead_bytes"] > 0 and st["resident_bytes"] == 0
        ctx.file_arena_release(l_f1)
        st = ctx.arena_cache_stats()
        assert st["leases"] == 0 and st["leased_dead_bytes"] == 0 and st["resident_bytes"] == 0 and st["resident_files"] == 0
        with pytest.raises(_lib.BloomGpuError):
            ctx.query([a3], [len(union)], Q.compile_queries(exprs[:1]))
        ctx.sync()
        assert abs(device_free_bytes() - free0) <= (8 << 20), "device memory did not come back"


def test_bad_arguments_fail_before_anything_changes():
    rng = np.random.default_rng(7)
    secs, *_ = make_file(rng, 4, max_tokens=50)
    with Context((0,)) as ctx:
        with pytest.raises(_lib.BloomGpuError, match="strictly ascending"):
            ctx.file_arena_acquire(b"k", [5, 5])
        aid, status = ctx.arena_load_sections(secs)
        with pytest.raises(_lib.BloomGpuError, match="strictly ascending"):
            ctx.file_arena_publish(b"k", aid, [4, 3, 2, 1], [0] * 4, [1] * 4, status)
        with pytest.raises(_lib.BloomGpuError, match="4 blocks, 3 keys"):
            ctx.file_arena_publish(b"k", aid, [1, 2, 3], [0] * 3, [1] * 3, status[:3])
        with pytest.raises(_lib.BloomGpuError):
            ctx.file_arena_publish(b"k", 987654, [1, 2, 3, 4], [0] * 4, [1] * 4, status)
        lease, resident = ctx.file_arena_publish(b"k", aid, [1, 2, 3, 4], [0] * 4, [1] * 4, status)
        assert resident
        with pytest.raises(_lib.BloomGpuError, match="already the cache's"):
            ctx.file_arena_publish(b"k2", aid, [1, 2, 3, 4], [0] * 4, [1] * 4, status)
        ctx.file_arena_release(lease)
        ctx.file_arena_forget(b"never seen")                         # forgetting an unknown file is not an error (TombstoneFile is idempotent)
        assert ctx.arena_cache_stats()["resident_files"] == 1
    # closing the context frees what the cache still held


@pytest.mark.parametrize("threads,forget_every", [(64, 0), (64, 23)])
def test_64_native_threads_over_a_budget_that_holds_a_quarter_of_the_files(threads, forget_every):
    """64 native threads (tools/native/conc_driver.cpp::cache_run) x random files x random candidate subsets; the budget holds about
    a quarter of the files, so arenas are evicted, re-read, widened and (second run) forgotten while leased, all at once.  Every
    candidate's verdict of every call is compared with the tree oracle's inside the driver."""
    from bloomsearch_amd import conc
    rng = np.random.default_rng(2026)
    n_files, nb = 16, 48
    files, expected = [], []
    made = [make_file(rng, nb, max_tokens=800) for _ in range(n_files)]
    vocab = made[0][4]
    exprs = [Q.And(Q.Token(vocab[int(rng.integers(0, 400))]), Q.Or(Q.Token(vocab[int(rng.integers(0, 5000))]), Q.Field("f%d" % rng.integers(0, 40))))
             for _ in range(12)] + [Q.Token(vocab[int(i)]) for i in rng.integers(0, 5000, size=12)] + [Q.Token("absent"), None]
    for secs, words, desc, _, _ in made:
        files.append(secs)
        expected.append(expected_survivors(words, desc, exprs))
    assert any(e.any() for e in expected) and not all(bool((e == e[0]).all()) for e in expected)
    with Context((0,)) as ctx:
        # size of one whole-file arena -> a budget of a quarter of the files
        lease, _, _ = load_and_publish(ctx, b"probe", files[0], range(nb))
        one = ctx.arena_cache_stats()["resident_bytes"]
        ctx.file_arena_release(lease)
        ctx.file_arena_forget(b"probe")
        ctx.set_arena_budget(one * n_files // 4)
        # a first, short run brings the library's scratch pool (regions, slot tables, staging: what `threads` concurrent loads hold at
        # once, kept for reuse) to its steady state; the leak check is that the long run after it leaves no more device memory behind
        warm = conc.cache_run(ctx, exprs, files, expected, n_threads=threads, seconds=0.7, forget_every=forget_every, seed=5)
        assert warm["errors"] == 0 and warm["mismatches"] == 0
        for f in range(n_files):
            ctx.file_arena_forget(bytes([f, 0, 0, 0]))
        ctx.sync()
        free0 = device_free_bytes()
        ctx.arena_cache_stats(reset=True)
        r = conc.cache_run(ctx, exprs, files, expected, n_threads=threads, seconds=3.0, forget_every=forget_every, seed=11)
        st = ctx.arena_cache_stats()
        assert r["errors"] == 0 and r["mismatches"] == 0, (r, st)
        assert r["calls"] > 20 * threads and r["hits"] > 0 and r["misses"] > 0
        assert st["hits"] == r["hits"] and st["misses"] == r["misses"]
        assert st["evictions"] > 0 and st["widenings"] > 0
        if forget_every:
            assert r["forgets"] > 0 and st["forgotten"] > 0
        # nothing is leased any more, nothing dead lingers, and the table respects its budget
        assert st["leases"] == 0 and st["leased_dead_bytes"] == 0
        assert st["resident_bytes"] <= st["budget_bytes"] and 0 < st["resident_files"] <= n_files
        for f in range(n_files):
            ctx.file_arena_forget(bytes([f, 0, 0, 0]))
        st = ctx.arena_cache_stats()
        assert st["resident_bytes"] == 0 and st["resident_files"] == 0
        ctx.sync()
        delta = free0 - device_free_bytes()
        assert delta <= (16 << 20), "device memory did not come back: %.1f MB more held than before the run (%d calls, %d arenas published)" % (
            delta / 1e6, r["calls"], st["published"])
