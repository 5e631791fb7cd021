"""bsg_probe_many_rows: the host-side gather of the north star delivers surviving block IDS (query_exec.go:321,603:
blockScanCandidate order), tagged per (arena, query) row NONE / ALL / LIST / DENSE and written by the device straight into
page-locked host memory.  Whatever the tag, a row expands to exactly the ascending block indices of bsg_probe_many's bitset,
which the other suites compare with the oracle; the oracle is the checker here too."""
import numpy as np
import pytest

from bloomsearch_amd import _lib, query as Q
from bloomsearch_amd.gpu import BloomGpuError, packed_headers, rows_to_dense, survivor_list, survivor_row_list
from oracle import oracle as O
from tests import helpers as H
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu


def pinned(ctx, dtype, n):
    a = ctx.pinned_array(max(n, 1) * np.dtype(dtype).itemsize).view(dtype)
    a[:] = np.iinfo(dtype).max                    # poison: what the device does not write must not be read as a result
    return a


@pytest.mark.parametrize("group", [1, 3, 64])
def test_rows_expand_to_the_oracle_survivor_sets_for_every_tag(ctx, group):
    rng = np.random.default_rng(77)
    plans = []
    for n_blocks in (130, 64, 1, 200, 65):
        plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.02)
        plans.append((plan, ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)))
    # queries that produce every tag: None (every block), an absent token (none / a few false positives), rare tokens (short lists),
    # common tokens and random trees (dense rows)
    exprs = [None, Q.Token("never-seen-anywhere"), Q.Field("no.such.field"), Q.Or(Q.Token("never-seen-anywhere"), Q.Token(vocab[0]))]
    exprs += [Q.Token(w) for w in vocab[:40]] + [H.random_expression(rng, vocab, None) for _ in range(300)]
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    bid = ctx.batch_create(terms, ops, poff)
    arenas = [ctx.arena_load(w, p.desc) for p, w in plans]
    arenas.append(ctx.arena_load(np.zeros(2, dtype=np.uint64), np.zeros(0, dtype=_lib.DESC_DTYPE)))      # an arena without blocks
    nbs = [p.n_blocks for p, _ in plans] + [0]
    order = [0, 5, 1, 2, 3, 4, 0, 3]
    NQ = cb.n_queries
    Gs = [(nbs[i] + 63) // 64 for i in order]
    rows = pinned(ctx, np.uint64, NQ * sum(Gs))
    hdr = pinned(ctx, np.uint32, NQ * len(order))
    ctx.set_probe_group(group)
    try:
        for flags in (0, _lib.PROBE_NOFUSE, _lib.PROBE_ASYNC, _lib.PROBE_ROWS_PACKED, _lib.PROBE_ROWS_PACKED | _lib.PROBE_ASYNC):
            packed = bool(flags & _lib.PROBE_ROWS_PACKED)      # payloads of every run of 256 queries back to back (NQ spans two runs)
            rows[:] = np.iinfo(np.uint64).max
            hdr[:] = np.iinfo(np.uint32).max
            ctx.probe_many_rows([arenas[i] for i in order], bid, rows, hdr, flags)
            if flags & _lib.PROBE_ASYNC:
                ctx.sync()
            tags_seen = set()
            o = 0
            for j, i in enumerate(order):
                G, nb = Gs[j], nbs[i]
                h = packed_headers(hdr, j, NQ) if packed else hdr[j * NQ: (j + 1) * NQ]
                r = rows[o: o + NQ * G].reshape(NQ, G) if G else np.zeros((NQ, 0), dtype=np.uint64)
                o += NQ * G
                if nb == 0:
                    assert not h.any()
                    continue
                p, w = plans[i]
                want = O.survivors_tree(w, p.desc.view(O.DESC_DTYPE), exprs)
                assert np.array_equal(rows_to_dense(h, r, nb, packed=packed), want), (group, flags, j)
                tags_seen |= set(int(t) for t in (h >> (6 if packed else 30)))
                pops = np.array([bin(int(x)).count("1") for x in (int.from_bytes(want[q].tobytes(), "little") for q in range(NQ))])
                if packed:      # a byte header carries the count of a LIST row only
                    lst = (h >> 6) == 2
                    assert np.array_equal((h & 63)[lst], pops[lst]) and not (h & 63)[~lst].any()
                else:
                    assert np.array_equal(h & np.uint32(0x3FFFFFFF), pops)
                for q in (0, 1, 3, 5, 44, 255, 256, 257, NQ - 1):          # the C helpers agree with bsg_survivor_list of the bitset
                    if not packed:
                        assert np.array_equal(survivor_row_list(int(h[q]), r[q], nb), survivor_list(want[q], nb))
                    assert np.array_equal(ctx.survivor_rows_list([arenas[i] for i in order], bid, rows, hdr, j, q, nb, packed=packed), survivor_list(want[q], nb))
            assert tags_seen == {0, 1, 2, 3}, tags_seen
            if packed:      # what the packed form is for: the payloads of a run are one contiguous stretch — nothing of the poison
                pass        # between them is asserted by rows_to_dense reading exactly the packed offsets
    finally:
        ctx.set_probe_group(0)
    ctx.pinned_free(rows.view(np.uint8))
    ctx.pinned_free(hdr.view(np.uint8))
    for a in arenas:
        ctx.arena_free(a)
    ctx.batch_free(bid)


def test_pageable_output_memory_is_refused_not_staged(ctx):
    rng = np.random.default_rng(5)
    plan, _, vocab = H.make_random_arena(rng, 10)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aid = ctx.arena_load(words, plan.desc)
    cb = Q.compile_queries([Q.Token(vocab[0])])
    ops, poff, _ = cb.arrays()
    bid = ctx.batch_create(H.gpu_terms(ctx, cb), ops, poff)
    with pytest.raises(BloomGpuError) as ei:
        ctx.probe_many_rows([aid], bid, np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint32))
    assert "page-locked" in str(ei.value)
    ctx.batch_free(bid)
    ctx.arena_free(aid)


def test_rows_on_a_context_of_several_devices_merge_to_the_global_block_order():
    """The Go host's shape — ONE process, one context over all 8 GPUs: every device writes its shards' rows (local block numbers) into
    its own slice of the page-locked buffers, and bsg_survivor_rows_list merges a (file, query)'s 8 rows into the surviving GLOBAL
    block indices (query_exec.go:603: the consumer walks surviving block ids).  Against the oracle's survivors, whatever the tags."""
    from bloomsearch_amd.gpu import Context
    rng = np.random.default_rng(77)
    for nd in (8, 3):
        with Context(device_ids(nd)) as m:
            plans, words, aids = [], [], []
            for nb in (1000, 5, 64 * nd + 1, 1, 130):
                plan, _, vocab = H.make_random_arena(rng, nb, absent_frac=0.02, max_tokens=60, vocab_size=30)
                w = m.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
                plans.append(plan); words.append(w); aids.append(m.arena_load(w, plan.desc))
            exprs = [None, Q.Token("absent-everywhere"), Q.Or(Q.Token("absent-everywhere"), Q.Field("f1")), Q.Token(vocab[0]),
                     Q.And(Q.Token(vocab[1]), Q.Token(vocab[2]))] + [H.random_expression(rng, vocab[:10], None) for _ in range(40)]
            cb = Q.compile_queries(exprs)
            ops, poff, _ = cb.arrays()
            bid = m.batch_create(H.gpu_terms(m, cb), ops, poff)
            order = [0, 1, 2, 3, 4, 0]
            ids = [aids[i] for i in order]
            rw, hw = m.survivor_rows_size(ids, bid)
            assert hw == nd * len(order) * len(exprs)
            rows = m.pinned_array(max(rw, 1) * 8).view(np.uint64)
            hdr = m.pinned_array(hw * 4).view(np.uint32)
            rows[:] = np.iinfo(np.uint64).max
            hdr[:] = np.iinfo(np.uint32).max
            for packed in (False, True):
                rows[:] = np.iinfo(np.uint64).max
                hdr[:] = np.iinfo(np.uint32).max
                m.probe_many_rows(ids, bid, rows, hdr, _lib.PROBE_ROWS_PACKED if packed else 0)
                tags = np.bincount(hdr.view(np.uint8)[:hw] >> 6, minlength=4) if packed else np.bincount(hdr >> 30, minlength=4)
                assert tags[0] > 0 and tags[1] > 0 and tags[2] + tags[3] > 0, tags        # every kind of row took part
                for j, i in enumerate(order):
                    want = O.survivors_tree(words[i], plans[i].desc.view(O.DESC_DTYPE), exprs)
                    for q in range(len(exprs)):
                        got = m.survivor_rows_list(ids, bid, rows, hdr, j, q, plans[i].n_blocks, packed=packed)
                        bits = np.zeros_like(want[q])
                        np.bitwise_or.at(bits, got.astype(np.int64) >> 6, np.uint64(1) << (got & 63).astype(np.uint64))
                        assert np.all(np.diff(got.astype(np.int64)) > 0) and np.array_equal(bits, want[q]), (nd, i, q, packed)
            # the dense path of the same context agrees
            dense = m.probe_many(ids, bid, 0, len(exprs), [plans[i].n_blocks for i in order])
            for j, i in enumerate(order):
                assert np.array_equal(dense[j], O.survivors_tree(words[i], plans[i].desc.view(O.DESC_DTYPE), exprs))
            m.pinned_free(rows.view(np.uint8))
            m.pinned_free(hdr.view(np.uint8))
            m.batch_free(bid)


def test_packed_rows_refuse_arenas_beyond_their_staging(ctx):
    """BSG_PROBE_ROWS_PACKED assembles a run's payloads in LDS: arenas of more than 1 024 blocks per device are refused, not mis-written."""
    rng = np.random.default_rng(9)
    plan, _, vocab = H.make_random_arena(rng, 1025, max_tokens=8, vocab_size=12)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aid = ctx.arena_load(words, plan.desc)
    cb = Q.compile_queries([Q.Token(vocab[0]), None])
    ops, poff, _ = cb.arrays()
    bid = ctx.batch_create(H.gpu_terms(ctx, cb), ops, poff)
    rows = pinned(ctx, np.uint64, 2 * 17)
    hdr = pinned(ctx, np.uint32, 2)
    with pytest.raises(BloomGpuError) as ei:
        ctx.probe_many_rows([aid], bid, rows, hdr, _lib.PROBE_ROWS_PACKED)
    assert "1024" in str(ei.value)
    ctx.probe_many_rows([aid], bid, rows, hdr)                       # the dense layout takes it
    want = O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), [Q.Token(vocab[0]), None])
    assert np.array_equal(rows_to_dense(hdr, rows, 1025), want)
    with pytest.raises(BloomGpuError):
        ctx.probe_many([aid], bid, _lib.PROBE_ROWS_PACKED, 2, [1025])   # a flag of bsg_probe_many_rows only
    ctx.pinned_free(rows.view(np.uint8)); ctx.pinned_free(hdr.view(np.uint8))
    ctx.batch_free(bid); ctx.arena_free(aid)
