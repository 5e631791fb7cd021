"""bsg_probe_many_rows: the host-side gather of the north star delivers surviving block IDS (query_exec.go:321,603:
blockScanCandidate order), tagged per (arena, query) row NONE / ALL / LIST / DENSE and written by the device straight into
page-locked host memory.  Whatever the tag, a row expands to exactly the ascending block indices of bsg_probe_many's bitset,
which the other suites compare with the oracle; the oracle is the checker here too."""
import numpy as np
import pytest

from bloomsearch_amd import _lib, query as Q
from bloomsearch_amd.gpu import BloomGpuError, rows_to_dense, survivor_list, survivor_row_list
from oracle import oracle as O
from tests import helpers as H

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
        for flags in (0, _lib.PROBE_NOFUSE, _lib.PROBE_ASYNC):
            rows[:] = np.iinfo(np.uint64).max
            hdr[:] = np.iinfo(np.uint32).max
            ctx.probe_many_rows([arenas[i] for i in order], bid, rows, hdr, flags)
            if flags & _lib.PROBE_ASYNC:
                ctx.sync()
            tags_seen = set()
            o = 0
            for j, i in enumerate(order):
                G, nb = Gs[j], nbs[i]
                h = hdr[j * NQ: (j + 1) * NQ]
                r = rows[o: o + NQ * G].reshape(NQ, G) if G else np.zeros((NQ, 0), dtype=np.uint64)
                o += NQ * G
                if nb == 0:
                    assert not h.any()
                    continue
                p, w = plans[i]
                want = O.survivors_tree(w, p.desc.view(O.DESC_DTYPE), exprs)
                assert np.array_equal(rows_to_dense(h, r, nb), want), (group, flags, j)
                tags_seen |= set(int(t) for t in (h >> 30))
                cnt = h & np.uint32(0x3FFFFFFF)
                assert np.array_equal(cnt, [bin(int(x)).count("1") for x in (int.from_bytes(want[q].tobytes(), "little") for q in range(NQ))])
                for q in (0, 1, 3, 5, 44, NQ - 1):                         # the C helper agrees with bsg_survivor_list of the bitset
                    assert np.array_equal(survivor_row_list(int(h[q]), r[q], nb), survivor_list(want[q], nb))
            assert tags_seen == {0, 1, 2, 3}, tags_seen
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
