"""Filters of 2^31 bits and more: every kernel selects its 64-bit-modulo instantiation by `d.m < (1ull << 31)`
(probe_block<false, ...>, build_entries<false>, build_from_slots<false>, the 64-bit mod_m of kernels.hip.h), and none of
those had met the oracle before this file.  A file-level filter crosses 2^31 bits at ~149 M distinct entries at p = 0.001
(one order above BASELINE configs[3]'s files), so this is product territory, not a corner.

Two geometries — m = 2^31 + 12 345 (268 MB per bitset) and m = 2^33 + 7 (1 GiB) — through every route that takes an m:
  build       bsg_build (entries), bsg_build_hashed, bsg_ingest_build (table slots -> k_build_sets)
  probe       few terms (k_probe_terms, gathered), > 128 terms (k_probe_terms_many), one interactive query (k_probe_direct)
  OR-reduce   bsg_or_reduce of two blocks of the same geometry == the oracle's build of the union
  wire        bsg_build_sections bytes == the oracle's encodeFilterSection; bsg_arena_load_sections decodes them back
Checker: oracle/bloom_oracle.c (bo_build_many, bo_encode_filter_section) and the tree-walking evaluator."""
import numpy as np
import pytest

from bloomsearch_amd import query as Q
from bloomsearch_amd._lib import DESC_DTYPE
from bloomsearch_amd.gpu import pack_entries
from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

K = 10
N_TOK = 20000


def tokens_of(block):
    return ["b%dtok%d" % (block, i) for i in range(N_TOK)] + ["shared%d" % i for i in range(500)]


def big_plan(m, n_blocks):
    """n_blocks blocks: field filter small, TOKEN filter of m bits, field::token filter small.  -> (blob, off, fstart, desc, n_words, entry lists)"""
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    entries, fstart, per_filter = [], [0], []
    cursor = 0
    for b in range(n_blocks):
        toks = tokens_of(b)
        sets = (["msg", "lvl"], toks, ["msg::" + t for t in toks[:300]])
        for c, s in enumerate(sets):
            mm, kk = (m, K) if c == 1 else O.estimate_parameters(len(s), 0.001)
            desc[b * 3 + c] = (cursor, mm, kk, 0)
            cursor += ((mm + 63) // 64 + 15) // 16 * 16
            entries += s
            per_filter.append(s)
            fstart.append(len(entries))
    blob, off = pack_entries(entries)
    return blob, off, np.asarray(fstart, dtype=np.uint32), desc, cursor, per_filter


@pytest.mark.parametrize("m", [(1 << 31) + 12345, (1 << 33) + 7], ids=["m=2^31+12345", "m=2^33+7"])
def test_64bit_modulo_build_probe_or_and_wire(ctx, m):
    n_blocks = 2
    blob, off, fstart, desc, n_words, per_filter = big_plan(m, n_blocks)
    odesc = desc.view(O.DESC_DTYPE)
    want = O.build_many(blob, off, fstart, odesc, n_words, n_threads=4)
    nw = O.words_for(m)
    # ---- build: entries route and hashed route (build_entries<false>: sliced global atomics; binning stops below 2^31) ----
    got = ctx.build(blob, off, fstart, desc, n_words)
    assert np.array_equal(got, want), "bsg_build differs from the oracle at m = %d" % m
    hashed = ctx.build_hashed(ctx.hash_entries(blob, off), fstart, desc, n_words)
    assert np.array_equal(hashed, want), "bsg_build_hashed differs from the oracle at m = %d" % m
    del hashed
    pop = int(np.unpackbits(want[int(desc[1]["word_off"]): int(desc[1]["word_off"]) + nw].view(np.uint8)).sum())
    assert 0.95 * K * len(per_filter[1]) < pop <= K * len(per_filter[1])      # the locations really spread over all of m

    # ---- probe ----
    members = [per_filter[1][i] for i in (0, 1, 777, N_TOK - 1)] + [per_filter[4][5], "shared7"]
    few = [Q.Token(t) for t in members] + [Q.Token("absent%d" % i) for i in range(40)] + \
          [Q.And(Q.Field("msg"), Q.Token("shared3"), Q.FieldToken("msg", per_filter[1][9])),
           Q.Or(Q.Token("nope"), Q.FieldToken("msg", "nope"), Q.Token(per_filter[4][0])), None]
    many = [Q.Token(t) for t in per_filter[1][:150]] + [Q.Token("missing%d" % i) for i in range(150)] + \
           [Q.Token(t) for t in per_filter[4][100:140]]
    one = [Q.And(Q.Token(per_filter[1][3]), Q.Token("shared1"), Q.Field("lvl"))]
    aid = ctx.arena_load(got, desc)
    try:
        for name, exprs in (("few-term", few), ("many-term", many), ("one query", one)):
            cb = Q.compile_queries(exprs)
            ops, poff, _ = cb.arrays()
            terms = H.gpu_terms(ctx, cb)
            w = O.survivors_tree(want, odesc, exprs)
            assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), w), name
            if name != "many-term":
                try:                                 # the same batch through the streaming kernels (k_probe_direct off)
                    ctx.set_lab(3, 0)
                    assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), w), name + " (k_probe_terms)"
                finally:
                    ctx.set_lab(3, 16)
            bits = [(int(x) >> b) & 1 for x in w[:, 0] for b in range(n_blocks)]
            assert any(bits) and (name == "one query" or not all(bits))        # hits and misses both occur
        # ---- fixed-geometry OR of the two token filters == build of the union at (m, K) ----
        union = sorted(set(per_filter[1]) | set(per_filter[4]))
        ub, uo = pack_entries(union)
        ud = np.zeros(1, dtype=O.DESC_DTYPE)
        ud[0] = (0, m, K, 0)
        want_or = O.build_many(ub, uo, np.asarray([0, len(union)], dtype=np.uint32), ud, nw, n_threads=1)
        assert np.array_equal(ctx.or_reduce(aid, 1, nw), want_or)
        del want_or
    finally:
        ctx.arena_free(aid)
    del got

    # ---- wire: sections written on the device, byte for byte the oracle's; and decoded back by the device ----
    secs = ctx.build_sections(blob, off, fstart, desc, n_words)
    for b in range(n_blocks):
        fl = []
        for c in range(3):
            d = desc[b * 3 + c]
            fl.append(O.Filter(int(d["m"]), int(d["k"]), want[int(d["word_off"]): int(d["word_off"]) + O.words_for(int(d["m"]))]))
        assert secs[b] == O.encode_filter_section(fl), "section %d" % b
        del fl
        if O.hw_crc32c_fn() is not None:      # the device's checksum of a section of up to 1 GiB vs the CPU's crc32 instruction
            sec = np.frombuffer(secs[b], dtype=np.uint8)
            assert int(sec[-4:].view("<u4")[0]) == O.hw_crc32c(sec[:-4]), "CRC32C of section %d" % b
    aid2, status = ctx.arena_load_sections(secs)
    del secs
    try:
        assert not status.any()
        cb = Q.compile_queries(few)
        ops, poff, _ = cb.arrays()
        assert np.array_equal(ctx.probe(aid2, n_blocks, H.gpu_terms(ctx, cb), ops, poff), O.survivors_tree(want, odesc, few))
    finally:
        ctx.arena_free(aid2)


def test_64bit_modulo_ingest_build_from_table_slots(ctx):
    """bsg_ingest_build with a caller-chosen geometry of 2^31 + 12 345 bits: k_build_sets -> build_from_slots<false>."""
    from oracle import walker_oracle as W
    m = (1 << 31) + 12345
    rows = [('{"msg":"%s","lvl":"L%d"}' % (" ".join("w%dx%d" % (r, j) for j in range(6)), r % 5)).encode() for r in range(3000)]
    sets = (set(), set(), set())
    for r in rows:
        W.index_row(r, sets)
    ing = ctx.ingest_rows(rows, [0, len(rows)], flags=1)
    try:
        assert len(ctx.ingest_fallback_rows(ing)) == 0
        counts, status = ctx.ingest_finish(ing, 1)
        assert [int(x) for x in counts[0]] == [len(s) for s in sets] and not status.any()
        desc = np.zeros(3, dtype=DESC_DTYPE)
        cursor = 0
        for c in range(3):
            mm, kk = (m, K) if c != 0 else O.estimate_parameters(len(sets[0]), 0.001)     # token AND field::token filters large
            desc[c] = (cursor, mm, kk, 0)
            cursor += ((mm + 63) // 64 + 15) // 16 * 16
        words = ctx.ingest_build(ing, desc, cursor)
    finally:
        ctx.ingest_free(ing)
    for c in range(3):
        ents = sorted(sets[c])
        b, o = pack_entries(ents)
        d1 = np.zeros(1, dtype=O.DESC_DTYPE)
        d1[0] = (0, int(desc[c]["m"]), int(desc[c]["k"]), 0)
        nw = O.words_for(int(desc[c]["m"]))
        want = O.build_many(b, o, np.asarray([0, len(ents)], dtype=np.uint32), d1, nw)
        assert np.array_equal(words[int(desc[c]["word_off"]): int(desc[c]["word_off"]) + nw], want), c


@pytest.mark.parametrize("m", [(1 << 30) + 77, (1 << 31) - 1], ids=["m=2^30+77", "m=2^31-1"])
def test_build_modulo_at_the_top_of_the_32_bit_range(ctx, m):
    """The 32-bit modulo of the build kernels (x - q m taken modulo 2^32, remainder candidate in [0, 2m)) at the top of its range —
    2m just below 2^32 — through the entries route (global atomics), the hashed route and the binned route (bsg_set_lab key 6 = 0:
    every bitset beyond LDS is binned), against the oracle's bitsets."""
    n_blocks = 1
    blob, off, fstart, desc, n_words, per_filter = big_plan(m, n_blocks)
    want = O.build_many(blob, off, fstart, desc.view(O.DESC_DTYPE), n_words, n_threads=4)
    assert np.array_equal(ctx.build(blob, off, fstart, desc, n_words), want), "bsg_build (global atomics) at m = %d" % m
    h = ctx.hash_entries(blob, off)
    assert np.array_equal(ctx.build_hashed(h, fstart, desc, n_words), want), "bsg_build_hashed at m = %d" % m
    try:
        ctx.set_lab(6, 0)
        assert np.array_equal(ctx.build(blob, off, fstart, desc, n_words), want), "binned bsg_build at m = %d" % m
    finally:
        ctx.set_lab(6, 4 << 20)
