"""BASELINE configs[1] at FULL size on the GPU (10 M rows / 1 000 blocks, Q = 4 096 three-term AND queries),
checked through size-independent properties plus an oracle check on a bounded sample:
  * bitsets: oracle rebuild of a sample of blocks is bit-identical; the XOR checksum over ALL words is stable
    across two independent GPU builds (build is a pure function of the entry sets);
  * no false negatives: a query whose three terms are all really present in block b survives in b;
  * monotonicity: survivors(And(a,b,c)) == survivors(a) & survivors(b) & survivors(c), Or likewise (bitwise);
  * idempotence: probing the same batch twice (and through the pipelined bsg_probe_many) gives identical bits;
  * a sample of queries is compared bit-for-bit with the oracle over all 1 000 blocks.
"""
import os

import numpy as np
import pytest

from bloomsearch_amd import query as Q, synth
from bloomsearch_amd._lib import TERM_DTYPE
from bloomsearch_amd.arena import plan_blocks
from oracle import oracle as O
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu

B, ROWS, NQ = 1000, 10000, 4096


def _gen(b):
    return synth.block_entry_sets(b * ROWS, ROWS)


@pytest.fixture(scope="module")
def c2(ctx):
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
        blocks = pool.map(_gen, range(B), chunksize=8)
    plan = plan_blocks(blocks, 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    return plan, words


def test_full_size_bitsets(ctx, c2):
    plan, words = c2
    again = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    assert np.bitwise_xor.reduce(words) == np.bitwise_xor.reduce(again) and np.array_equal(words, again)
    # sizing: exact distinct counts => EstimateParameters(count, 0.001) for every filter
    for b in (0, 499, 999):
        for c in range(3):
            assert (int(plan.desc["m"][b * 3 + c]), int(plan.desc["k"][b * 3 + c])) == O.estimate_parameters(max(int(plan.counts[b, c]), 1), 0.001)
    # oracle rebuild of a bounded sample of filters, bit for bit
    for f in (0, 1, 2, 3 * 500 + 1, 3 * 999 + 2):
        d = plan.desc[f]
        e0, e1 = int(plan.fstart[f]), int(plan.fstart[f + 1])
        filt = O.Filter(int(d["m"]), int(d["k"]))
        raw = plan.blob.tobytes()
        for e in range(e0, e1):
            filt.add(raw[int(plan.off[e]): int(plan.off[e + 1])])
        nw = O.words_for(int(d["m"]))
        assert np.array_equal(filt.words, words[int(d["word_off"]): int(d["word_off"]) + nw])


def test_full_size_probe_properties(ctx, c2):
    plan, words = c2
    rng = np.random.default_rng(2)
    d = synth.draws(0, B * ROWS)
    # queries built from real rows: row r of block b has (level, service, region) => the AND must survive in block b
    rows = rng.integers(0, B * ROWS, size=NQ // 2)
    exprs, must = [], []
    for r in rows:
        lv, sv, rg = synth.LEVELS[d["level"][r]], synth.SERVICES[d["service"][r]], "region-%d" % d["region"][r]
        uid = str(int(d["user_id"][r]))
        exprs.append(Q.And(Q.FieldToken("level", lv), Q.FieldToken("service", sv), Q.FieldToken("user_id", uid)))
        must.append(int(r) // ROWS)
    for _ in range(NQ - len(exprs)):
        exprs.append(Q.And(Q.FieldToken("level", synth.LEVELS[rng.integers(0, 4)]), Q.FieldToken("service", "absent-%d" % rng.integers(0, 9)),
                           Q.FieldToken("nested.region", "region-%d" % rng.integers(0, 12))))
    cb = Q.compile_queries(exprs)
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    aid = ctx.arena_load(words, plan.desc)
    bid = ctx.batch_create(terms, ops, poff)
    got = ctx.probe_batch(aid, bid, NQ, B)
    # no false negatives
    for q, b in enumerate(must):
        assert (int(got[q, b >> 6]) >> (b & 63)) & 1, (q, b)
    # idempotence, also through the fused/pipelined path
    assert np.array_equal(got, ctx.probe_batch(aid, bid, NQ, B))
    many = ctx.probe_many([aid, aid, aid], bid, n_queries=NQ, n_blocks=[B, B, B])
    assert all(np.array_equal(m, got) for m in many)
    # monotonicity: And == bitwise AND of its single-term probes
    singles = Q.compile_queries([Q.FieldToken("level", "error"), Q.FieldToken("service", "payment"), Q.FieldToken("user_id", "4242"),
                                 Q.And(Q.FieldToken("level", "error"), Q.FieldToken("service", "payment"), Q.FieldToken("user_id", "4242")),
                                 Q.Or(Q.FieldToken("level", "error"), Q.FieldToken("service", "payment"), Q.FieldToken("user_id", "4242"))])
    sops, spoff, skinds = singles.arrays()
    sterms = np.zeros(len(singles.term_strings), dtype=TERM_DTYPE)
    sterms["h"] = ctx.hash_strings(singles.term_strings)
    sterms["kind"] = skinds
    s = ctx.probe(aid, B, sterms, sops, spoff)
    assert np.array_equal(s[3], s[0] & s[1] & s[2]) and np.array_equal(s[4], s[0] | s[1] | s[2])
    # bounded oracle sample: 48 queries over all 1 000 blocks, bit for bit
    idx = np.concatenate([np.arange(24), np.arange(NQ - 24, NQ)])
    sub = Q.compile_queries([exprs[i] for i in idx])
    o2, p2, k2 = sub.arrays()
    t2 = np.zeros(len(sub.term_strings), dtype=O.TERM_DTYPE)
    for i, sname in enumerate(sub.term_strings):
        t2["h"][i] = O.base_hashes(sname)
        t2["kind"][i] = k2[i]
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), t2, o2, p2)
    assert np.array_equal(got[idx], want)
    # the last block group is partial (1000 = 15 * 64 + 40): no bits beyond block 999
    assert not (got[:, -1] >> np.uint64(40)).any()
    ctx.batch_free(bid)
    ctx.arena_free(aid)


def _gen_rows(b):
    rs = synth.rows_json(b * ROWS, ROWS)
    return b"".join(rs), np.asarray([len(r) for r in rs], dtype=np.uint32)


def test_full_size_device_ingest_equals_entry_set_route(ctx, c2):
    """BASELINE configs[2] from the front of the path: the JSON rows of 40 full-size blocks (400 000 rows) through
    k_ingest_rows / k_ingest_union / k_build_sets.  Block filters must equal, bit for bit, the ones bsg_build made from
    the pre-extracted entry sets of the same blocks (which test_full_size_bitsets pins against the oracle); the
    file-level counts must equal the size of the union of the blocks' entry sets, and the file-level token filter must
    contain every token of a sampled block (no false negatives) at the union's geometry."""
    import multiprocessing as mp
    from bloomsearch_amd import ingest as I
    plan, words = c2
    nb = 40
    with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
        parts = pool.map(_gen_rows, range(nb))
    blob = np.frombuffer(b"".join(p[0] for p in parts), dtype=np.uint8)
    lens = np.concatenate([p[1] for p in parts])
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    first = np.arange(nb + 1, dtype=np.uint32) * ROWS
    for flags in (0, 1):
        ing = ctx.ingest_rows((blob, off), first, np.zeros(nb, dtype=np.uint32), 1, flags=flags)
        assert len(ctx.ingest_fallback_rows(ing)) == 0
        counts, status = ctx.ingest_finish(ing, nb + 1)
        assert not status.any()
        assert np.array_equal(counts[:nb].astype(np.int64), plan.counts[:nb])
        desc, n_words = I.plan_desc(counts, 0.001)
        got = ctx.ingest_build(ing, desc, n_words)
        ctx.ingest_free(ing)
        for i in range(nb * 3):
            d, e = desc[i], plan.desc[i]
            nw = (int(e["m"]) + 63) // 64
            assert (int(d["m"]), int(d["k"])) == (int(e["m"]), int(e["k"]))
            assert np.array_equal(got[int(d["word_off"]): int(d["word_off"]) + nw], words[int(e["word_off"]): int(e["word_off"]) + nw]), i
    # file level: exact union counts, right-sized geometry, members present
    raw = plan.blob.tobytes()
    union = [set(), set(), set()]
    for f in range(nb * 3):
        for e in range(int(plan.fstart[f]), int(plan.fstart[f + 1])):
            union[f % 3].add(raw[int(plan.off[e]): int(plan.off[e + 1])])
    assert [int(x) for x in counts[nb]] == [len(u) for u in union]
    for c in range(3):
        d = desc[nb * 3 + c]
        assert (int(d["m"]), int(d["k"])) == O.estimate_parameters(len(union[c]), 0.001)
    d = desc[nb * 3 + 1]
    filt = O.Filter(int(d["m"]), int(d["k"]), got[int(d["word_off"]): int(d["word_off"]) + O.words_for(int(d["m"]))])
    sample = sorted(union[1])[::997]
    assert all(filt.test(t) for t in sample)
    want = O.Filter(int(d["m"]), int(d["k"]))          # and the file-level field filter entirely (9 entries)
    df = desc[nb * 3]
    ff = O.Filter(int(df["m"]), int(df["k"]))
    for t in union[0]:
        ff.add(t)
    assert np.array_equal(ff.words, got[int(df["word_off"]): int(df["word_off"]) + O.words_for(int(df["m"]))])


def test_full_size_c4_eight_term_or_batch_single_and_eight_entry_context(ctx, c2):
    """BASELINE configs[3]'s query shape at FULL block size: one file of 1 000 blocks x 10 000 rows, Q = 4 096 eight-term
    Or(FieldToken...) queries (the bench's generator + needles built from real rows), on the single-device context and on a
    context of 8 entries (block b on entry b % 8, survivors interleaved on the host):
      * no false negatives: an Or holding one (field, value) of a real row survives in that row's block;
      * Or == bitwise OR of its eight single-term probes;
      * idempotence across launch groupings (1 / 3 arenas per dispatch, fused or not) and across the two contexts;
      * a RANDOM sample of 48 queries equals the tree-walking oracle over all 1 000 blocks, bit for bit."""
    from bloomsearch_amd.gpu import Context
    from tests import helpers as H
    plan, words = c2
    rng = np.random.default_rng(8)
    d = synth.draws(0, B * ROWS)
    exprs = synth.make_queries(NQ - 512, "c4", seed=99)
    must = []
    for r in rng.integers(0, B * ROWS, size=512):
        uid = str(int(d["user_id"][r]))
        absent = [Q.FieldToken("level", "absent-level-%d" % rng.integers(0, 4)), Q.FieldToken("service", "absent-svc-%d" % rng.integers(0, 4)),
                  Q.FieldToken("nested.region", "region-%d" % rng.integers(8, 12)), Q.FieldToken("nested.az", "az-%d" % rng.integers(3, 6)),
                  Q.FieldToken("tags", "absent-word-%d" % rng.integers(0, 8)), Q.FieldToken("message", "absent-word-%d" % rng.integers(0, 8)),
                  Q.FieldToken("user_id", "nobody-%d" % rng.integers(0, 99))]
        pos = int(rng.integers(0, 8))
        exprs.append(Q.Or(*(absent[:pos] + [Q.FieldToken("user_id", uid)] + absent[pos:])))
        must.append((len(exprs) - 1, int(r) // ROWS))
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    aid = ctx.arena_load(words, plan.desc)
    bid = ctx.batch_create(terms, ops, poff)
    got = ctx.probe_batch(aid, bid, NQ, B)
    for q, b in must:
        assert (int(got[q, b >> 6]) >> (b & 63)) & 1, (q, b)
    needle_rows = got[[q for q, _ in must]]
    survive = np.unpackbits(needle_rows.view(np.uint8), axis=1, bitorder="little")[:, :B].sum(axis=1)
    assert 60 < survive.mean() < 150                             # a user's events: ~100 rows in ~95 of the 1 000 blocks (+ ~7 false positives), not all of them
    # Or == bitwise OR of the single-term probes (first 40 queries)
    for q in range(0, 40):
        kids = exprs[q]["Children"]
        sub = Q.compile_queries(kids)
        so, sp, _ = sub.arrays()
        s = ctx.probe(aid, B, H.gpu_terms(ctx, sub), so, sp)
        assert np.array_equal(np.bitwise_or.reduce(s, axis=0), got[q]), q
    # idempotence across groupings
    try:
        for group in (1, 3):
            ctx.set_probe_group(group)
            many = ctx.probe_many([aid, aid, aid, aid], bid, 0, NQ, [B] * 4)
            assert all(np.array_equal(m, got) for m in many), group
    finally:
        ctx.set_probe_group(0)
    # random oracle sample
    sel = np.sort(rng.choice(NQ, size=48, replace=False))
    want = O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), [exprs[int(i)] for i in sel])
    assert np.array_equal(got[sel], want)
    assert not (got[:, -1] >> np.uint64(40)).any()
    ctx.batch_free(bid)
    ctx.arena_free(aid)
    # the same file and batch on a context of 8 entries
    with Context(device_ids(8)) as m8:
        a8 = m8.arena_load(words, plan.desc)
        b8 = m8.batch_create(terms, ops, poff)
        g8 = m8.probe_batch(a8, b8, NQ, B)
        assert np.array_equal(g8, got)
        many = m8.probe_many([a8, a8], b8, 0, NQ, [B, B])
        assert np.array_equal(many[0], got) and np.array_equal(many[1], got)
