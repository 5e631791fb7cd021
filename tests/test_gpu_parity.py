"""GPU parity tests proper: every result of the HIP path (through the C-ABI) is
compared bit-for-bit with the CPU oracle on the same seeded inputs.  Integer /
byte work => the bar is exact equality."""
import os

import numpy as np
import pytest

from bloomsearch_amd import query as Q
from bloomsearch_amd._lib import DESC_DTYPE, TERM_DTYPE, BloomGpuError, BSG_E_INVALID, op, OP_AND, OP_TERM
from bloomsearch_amd.arena import entry_sets_from_strings, plan_blocks
from bloomsearch_amd.gpu import pack_entries
from oracle import oracle as O
from tests import helpers as H
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu


def test_hash_entries_bit_exact(ctx):
    rng = np.random.default_rng(11)
    ents = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in list(range(0, 70)) * 3 + [255, 256, 257, 4095]]
    ents += [b"user.name", b"user.name::alice", "日本語::héllo".encode(), b"", b"\x00", b"\x01"]
    got = ctx.hash_strings(ents)
    want = np.array([O.base_hashes(e) for e in ents], dtype=np.uint64)
    assert np.array_equal(got, want)


def test_device_hashes_equal_the_independent_murmur3(ctx):
    """k_hash_entries against Austin Appleby's own MurmurHash3_x64_128 (oracle.appleby: scikit-learn's bundled MurmurHash3.cpp,
    compiled from where it lies) — not via the restatement: (h0, h1) = murmur(d), (h2, h3) = murmur(d + 0x01) for every length
    0..300 and a few thousand more bytes, so the HIP hash is pinned by an implementation written by neither side."""
    if O.appleby() is None:
        pytest.skip("scikit-learn's MurmurHash3.cpp is not in this image")
    rng = np.random.default_rng(77)
    ents = [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in range(0, 301) for _ in range(2)]
    ents += [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in rng.integers(301, 4000, size=100)]
    got = ctx.hash_strings(ents)
    for row, d in zip(got, ents):
        assert tuple(int(x) for x in row) == O.appleby_x64_128(d) + O.appleby_x64_128(d + b"\x01"), len(d)


def test_plain_c99_caller_hashes_on_the_gpu(tmp_path):
    """The C program of tests/test_cabi.py, where a GPU is present: bsg_open succeeds and bsg_hash_entries returns the
    public MurmurHash3_x64_128 vector for "hello" — the boundary works without Python or torch in the process."""
    from tests.test_cabi import run_c_caller
    out = run_c_caller(tmp_path)
    assert "open=0" in out and "hello=cbd8a7b341bd9b02 5b1e906a48ae1d19" in out
    # ... and the reference's own TestEvaluateBloomFilters fixture, run from C: the section bytes bsg_build_sections wrote are the
    # golden wire bytes, and the eight verdicts of ONE bsg_query call are the reference's (bloom_tree_engine_test.go:382-427)
    import json
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bloom_vectors.json")))["evaluate_bloom_filters_fixture"]
    lines = dict(l.split("=", 1) for l in out.splitlines() if "=" in l and " " not in l.split("=", 1)[0])
    assert lines["section"] == fx["section_hex"]
    assert lines["verdicts"] == "".join("1" if c["expected"] else "0" for c in fx["cases"]) == "11011101"


def test_hash_entries_large_batch(ctx):
    rng = np.random.default_rng(12)
    ents = [b"tok%d" % i for i in rng.integers(0, 1 << 40, size=50000)]
    got = ctx.hash_strings(ents)
    idx = rng.integers(0, len(ents), size=500)
    for i in idx:
        assert tuple(int(x) for x in got[i]) == O.base_hashes(ents[i])


def test_build_bit_exact_small_medium_large_empty_absent(ctx):
    rng = np.random.default_rng(13)
    blocks = [
        entry_sets_from_strings(["a", "b.c"], ["x%d" % i for i in range(300)], []),                # empty FT set => n=1, no bits
        entry_sets_from_strings(["f%d" % i for i in range(9)], ["t%d" % i for i in range(20000)],   # ~35 KB (LDS staged)
                                ["f::t%d" % i for i in range(20000)]),
        entry_sets_from_strings(["only"], ["big%d" % i for i in range(60000)], ["k::v"]),            # 105 KiB: still LDS-staged (opt-in LDS)
        entry_sets_from_strings(["only"], ["huge%d" % i for i in range(160000)], ["k::v"]),          # > 144 KiB budget at both fprs => global-atomic path
        entry_sets_from_strings([], [], []),
    ]
    for fpr in (0.001, 0.01):
        plan = plan_blocks(blocks, fpr, absent={(4, 1)})
        want = H.oracle_words(plan)
        # the bitset beyond LDS: binned locations assembled window by window (k_bin_*), then — lab knob 2 = 0 — the
        # global-atomic path filters of 2^31 bits and more still take; also from precomputed hashes
        try:
            ctx.set_lab(6, 0)                                  # (by default only bitsets with millions of locations are binned)
            got = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
            hashed = ctx.build_hashed(ctx.hash_entries(plan.blob, plan.off), plan.fstart, plan.desc, plan.n_words)
            ctx.set_lab(2, 0)
            atomics = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        finally:
            ctx.set_lab(2, 16 << 30)
            ctx.set_lab(6, 4 << 20)
        assert np.array_equal(got, want) and np.array_equal(hashed, want) and np.array_equal(atomics, want)
        assert np.array_equal(ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words), want)     # the defaults
        # sizing rule: exact distinct counts (TestMeasuredFilterSizing, file_format_test.go:28-94)
        assert plan.desc["m"][1 * 3 + 1] == O.estimate_parameters(20000, fpr)[0]
        assert plan.desc["m"][0 * 3 + 2] == O.estimate_parameters(1, fpr)[0]
        assert plan.desc["m"][4 * 3 + 1] == 0
        assert (int(plan.desc["m"][3 * 3 + 1]) + 63) // 64 * 8 > 144 * 1024


def test_build_sections_bytes_equal_oracle_and_host_codec(ctx):
    """encodeFilterSection on the device (k_encode_payload + k_crc_sections): every section is byte-identical to the
    oracle's encode of the oracle's filters and to the C++ host codec; absent filters clear their flag bit; the
    sections round-trip through the device decoder (bsg_arena_load_sections) with clean status."""
    from bloomsearch_amd import host as Hst
    blocks = [
        entry_sets_from_strings(["a", "b.c"], ["x%d" % i for i in range(300)], []),
        entry_sets_from_strings(["f%d" % i for i in range(9)], ["t%d" % i for i in range(20000)], ["f::t%d" % i for i in range(20000)]),
        entry_sets_from_strings(["only"], ["big%d" % i for i in range(60000)], ["k::v"]),
        entry_sets_from_strings(["only"], ["huge%d" % i for i in range(160000)], ["k::v"]),
        entry_sets_from_strings([], [], []),
        entry_sets_from_strings(["z"], ["one"], ["z::one"]),
    ]
    for fpr, absent in ((0.001, {(4, 1)}), (0.01, {(0, 0), (0, 1), (0, 2), (5, 2)})):
        plan = plan_blocks(blocks, fpr, absent=absent)
        secs = ctx.build_sections(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        assert ctx.last_encode_ms() > 0
        words = H.oracle_words(plan)
        assert ctx.sections_size(plan.desc) == sum(len(x) for x in secs)
        if O.hw_crc32c_fn() is not None:      # k_crc_sections against the CPU's own crc32 instruction (oracle/hw_crc32c.c)
            for x in secs:
                assert int(np.frombuffer(x[-4:], dtype="<u4")[0]) == O.hw_crc32c(x[:-4])
        for b in range(len(blocks)):
            fl_o, fl_h = [], []
            for c in range(3):
                d = plan.desc[b * 3 + c]
                if int(d["m"]) == 0:
                    fl_o.append(None); fl_h.append(None)
                    continue
                w = words[int(d["word_off"]): int(d["word_off"]) + O.words_for(int(d["m"]))]
                fl_o.append(O.Filter(int(d["m"]), int(d["k"]), w))
                fl_h.append((int(d["m"]), int(d["k"]), w))
            want = O.encode_filter_section(fl_o)
            assert secs[b] == want, (fpr, b)
            assert Hst.section_encode(fl_h) == want
        aid, status = ctx.arena_load_sections(secs)
        assert not status.any()
        ctx.arena_free(aid)


def test_build_hashed_equals_build(ctx):
    blocks = [entry_sets_from_strings(["p"], ["w%d" % i for i in range(5000)], ["p::w%d" % i for i in range(5000)])]
    plan = plan_blocks(blocks, 0.001)
    a = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    h = ctx.hash_entries(plan.blob, plan.off)
    b = ctx.build_hashed(h, plan.fstart, plan.desc, plan.n_words)
    assert np.array_equal(a, b)
    assert np.array_equal(a, H.oracle_words(plan))


def test_build_order_invariance_and_no_false_negatives(ctx):
    rng = np.random.default_rng(14)
    toks = ["tok%d" % i for i in range(4000)]
    p1 = plan_blocks([entry_sets_from_strings(["f"], toks, [])], 0.001)
    p2 = plan_blocks([entry_sets_from_strings(["f"], [toks[i] for i in rng.permutation(len(toks))], [])], 0.001)
    w1 = ctx.build(p1.blob, p1.off, p1.fstart, p1.desc, p1.n_words)
    w2 = ctx.build(p2.blob, p2.off, p2.fstart, p2.desc, p2.n_words)
    assert np.array_equal(w1, w2)
    d = p1.desc[1]
    f = O.Filter(int(d["m"]), int(d["k"]), w1[int(d["word_off"]): int(d["word_off"]) + O.words_for(int(d["m"]))])
    assert all(f.test(t) for t in toks)


@pytest.mark.parametrize("n_blocks,seed", [(1, 1), (63, 2), (64, 3), (65, 4), (130, 5), (257, 6)])
def test_probe_random_arena_random_trees(ctx, n_blocks, seed):
    rng = np.random.default_rng(100 + seed)
    plan, blocks_str, vocab = H.make_random_arena(rng, n_blocks)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    assert np.array_equal(words, H.oracle_words(plan))
    exprs = [None] + [H.random_expression(rng, vocab, None) for _ in range(300)]
    # make sure members are probed too: terms that really exist in some block
    for b in rng.integers(0, n_blocks, size=40):
        f, t, ft = blocks_str[b]
        if t:
            fld, tok = ft[rng.integers(0, len(ft))].split("::", 1)
            exprs.append(Q.And(Q.Field(f[0]), Q.Token(t[rng.integers(0, len(t))]), Q.FieldToken(fld, tok)))
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    assert np.array_equal(terms["h"], H.oracle_terms(cb)["h"])
    aid = ctx.arena_load(words, plan.desc)
    try:
        got = ctx.probe(aid, n_blocks, terms, ops, poff)
    finally:
        ctx.arena_free(aid)
    # the checker walks the expression TREES (oracle.evaluate_tree*, query_exec.go:89-159): nothing of the product's
    # lowering (compile_queries -> postfix) sits between the expressions and the expected survivor sets
    want = O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), exprs)
    assert np.array_equal(got, want)
    assert np.array_equal(want, O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff))
    assert want.any() and not want.all()


def test_probe_nil_filters_fail_open_and_constants(ctx):
    blocks = [entry_sets_from_strings(["a"], ["x"], ["a::x"]) for _ in range(5)]
    plan = plan_blocks(blocks, 0.01, absent={(1, 0), (2, 1), (3, 2), (4, 0), (4, 1), (4, 2)})
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    exprs = [Q.Field("zzz"), Q.Token("zzz"), Q.FieldToken("q", "zzz"), Q.And(), Q.Or(), None,
             Q.And(Q.Field("a"), Q.Token("x"), Q.FieldToken("a", "x")),
             {"ExpressionType": "CONDITION", "Condition": None}, {"ExpressionType": "NOPE"}]
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    aid = ctx.arena_load(words, plan.desc)
    got = ctx.probe(aid, 5, terms, ops, poff)
    ctx.arena_free(aid)
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(got, want)
    bits = lambda q: [(int(got[q, 0]) >> b) & 1 for b in range(5)]
    assert bits(0) == [0, 1, 0, 0, 1]      # nil field filter cannot disqualify (query_exec.go:137-140)
    assert bits(1) == [0, 0, 1, 0, 1]
    assert bits(2) == [0, 0, 0, 1, 1]
    assert bits(3) == [1] * 5 and bits(4) == [0] * 5 and bits(5) == [1] * 5
    assert bits(6) == [1] * 5 and bits(7) == [1] * 5 and bits(8) == [0] * 5


def test_probe_oversize_filter_takes_gather_path(ctx):
    big = ["big%d" % i for i in range(110000)]
    blocks = [entry_sets_from_strings(["f"], big, ["f::" + t for t in big[:100]]),
              entry_sets_from_strings(["f"], ["small"], ["f::small"])]
    plan = plan_blocks(blocks, 0.001)
    assert (int(plan.desc["m"][1]) + 63) // 64 * 8 > 144 * 1024
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    assert np.array_equal(words, H.oracle_words(plan))
    exprs = [Q.Token(t) for t in big[:200]] + [Q.Token("nope%d" % i) for i in range(200)] + [Q.Token("small")]
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    aid = ctx.arena_load(words, plan.desc)
    got = ctx.probe(aid, 2, terms, ops, poff)
    ctx.arena_free(aid)
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(got, want)
    assert all(int(got[q, 0]) & 1 for q in range(200))


def test_probe_resident_batch_reuse_and_timing(ctx):
    from bloomsearch_amd._lib import PROBE_TIMED
    rng = np.random.default_rng(21)
    plan, blocks_str, vocab = H.make_random_arena(rng, 100, absent_frac=0.0)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    cb = Q.compile_queries([H.random_expression(rng, vocab, None) for _ in range(700)])
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    aid = ctx.arena_load(words, plan.desc)
    bid = ctx.batch_create(terms, ops, poff)
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    ctx.timing_read(reset=True)
    for _ in range(3):
        got = ctx.probe_batch(aid, bid, cb.n_queries, 100, flags=PROBE_TIMED)
        assert np.array_equal(got, want)
    t = ctx.timing_read()
    assert t.n_probes == 3 and t.ms_terms_kernel > 0 and t.ms_eval_kernel > 0 and t.stream_bytes > 0
    ctx.batch_free(bid)
    ctx.arena_free(aid)


def test_malformed_inputs_rejected_before_launch(ctx):
    desc = np.zeros(3, dtype=DESC_DTYPE)
    desc[1] = (0, 6400, 3, 0)
    with pytest.raises(BloomGpuError) as e:        # words outside the arena
        ctx.arena_load(np.zeros(10, dtype=np.uint64), desc)
    assert e.value.code == BSG_E_INVALID
    aid = ctx.arena_load(np.zeros(100, dtype=np.uint64), desc)
    terms = np.zeros(1, dtype=TERM_DTYPE)
    with pytest.raises(BloomGpuError):             # TERM index out of range
        ctx.probe(aid, 1, terms, [op(OP_TERM, 5)], [0, 1])
    with pytest.raises(BloomGpuError):             # AND pops more than the stack holds
        ctx.probe(aid, 1, terms, [op(OP_TERM, 0), op(OP_AND, 2)], [0, 2])
    with pytest.raises(BloomGpuError):             # two values left on the stack
        ctx.probe(aid, 1, terms, [op(OP_TERM, 0), op(OP_TERM, 0)], [0, 2])
    terms["kind"] = 7
    with pytest.raises(BloomGpuError):
        ctx.probe(aid, 1, terms, [op(OP_TERM, 0)], [0, 1])
    ctx.arena_free(aid)
    with pytest.raises(BloomGpuError):
        ctx.arena_free(aid)


def test_or_reduce_fixed_geometry_equals_build_of_union(ctx):
    # OR_b build(S_b, m, k) == build(U S_b, m, k) under one geometry
    rng = np.random.default_rng(31)
    universe = ["tok%d" % i for i in range(30000)]
    m, k = O.estimate_parameters(len(universe), 0.001)
    n_blocks = 37
    parts = [[] for _ in range(n_blocks)]
    for t in universe:
        parts[rng.integers(0, n_blocks)].append(t)
    nw = O.words_for(m)
    stride = (nw + 1) // 2 * 2
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    fstart = [0]
    ents = []
    for b in range(n_blocks):
        fstart += [len(ents)]                                  # field: absent
        desc[b * 3 + 1] = (b * stride, m, k, 0)
        ents += parts[b]
        fstart += [len(ents), len(ents)]                       # ft: absent
    blob, off = pack_entries(ents)
    words = ctx.build(blob, off, np.asarray(fstart, dtype=np.uint32), desc, n_blocks * stride)
    aid = ctx.arena_load(words, desc)
    got = ctx.or_reduce(aid, 1, nw)
    ctx.arena_free(aid)
    want = O.Filter(m, k)
    for t in universe:
        want.add(t)
    assert np.array_equal(got, want.words)


def test_probe_many_pipelined_equals_individual_probes(ctx):
    """bsg_probe_many software-pipelines K2(i) beside K1(i+1) on double-buffered scratch: results must be
    those of one-at-a-time probes, for arenas of different shapes and odd/even counts."""
    rng = np.random.default_rng(77)
    arenas, wants, nbs = [], [], []
    vocab = None
    plans = []
    for n_blocks in (70, 3, 129, 64, 200):
        plan, blocks_str, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.02)
        plans.append((plan, ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)))
    cb = Q.compile_queries([None] + [H.random_expression(rng, vocab, None) for _ in range(600)])
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    bid = ctx.batch_create(terms, ops, poff)
    for plan, words in plans:
        arenas.append(ctx.arena_load(words, plan.desc))
        nbs.append(plan.n_blocks)
        wants.append(O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff))
    for ids in ([0, 1, 2, 3, 4], [4, 4, 0], [2], [1, 3, 1, 3, 1, 3, 0]):
        got = ctx.probe_many([arenas[i] for i in ids], bid, n_queries=cb.n_queries, n_blocks=[nbs[i] for i in ids])
        for g, i in zip(got, ids):
            assert np.array_equal(g, wants[i]), ids
    # async form + a following synchronous probe still sees consistent scratch
    ctx.probe_many([arenas[i] for i in (0, 1, 2, 3)], bid)
    assert np.array_equal(ctx.probe_batch(arenas[4], bid, cb.n_queries, nbs[4]), wants[4])
    for a in arenas:
        ctx.arena_free(a)
    ctx.batch_free(bid)


def test_large_block_filter_is_lds_staged_and_bit_exact(ctx):
    """A 10 MiB-row-group-sized block (70k distinct tokens => 123 KiB bitset) fits the opt-in LDS budget."""
    toks = ["tok%d" % i for i in range(70000)]
    plan = plan_blocks([entry_sets_from_strings(["f"], toks, ["f::x"]) for _ in range(3)], 0.001)
    nbytes = (int(plan.desc["m"][1]) + 63) // 64 * 8
    assert 64 * 1024 < nbytes < 144 * 1024
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    assert np.array_equal(words, H.oracle_words(plan))
    exprs = [Q.Token(t) for t in toks[:300]] + [Q.Token("nope%d" % i) for i in range(300)]
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    aid = ctx.arena_load(words, plan.desc)
    got = ctx.probe(aid, 3, terms, ops, poff)
    ctx.arena_free(aid)
    assert np.array_equal(got, O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff))


def test_multi_device_context_shards_round_robin_and_gathers():
    """A context over several devices shards blocks b -> device b % n and gathers survivor bitsets on the host.
    The GPU box has one GPU, so the same device is listed twice/thrice: the sharding, per-shard launches and the
    host-side interleave are exactly the N-GPU code path."""
    from bloomsearch_amd.gpu import Context
    rng = np.random.default_rng(55)
    for devs, n_blocks in ((device_ids(2), 131), (device_ids(3), 64), (device_ids(2), 1)):
        plan, blocks_str, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.03)
        with Context(devs) as mctx:
            words = mctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
            cb = Q.compile_queries([None] + [H.random_expression(rng, vocab, None) for _ in range(300)])
            ops, poff, _ = cb.arrays()
            terms = H.gpu_terms(mctx, cb)
            aid = mctx.arena_load(words, plan.desc)
            got = mctx.probe(aid, n_blocks, terms, ops, poff)
            want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
            assert np.array_equal(got, want)
            # fixed-geometry OR-reduce across the shards of a multi-device context
            mctx.arena_free(aid)


def test_arena_load_sections_device_decode(ctx):
    """Filter sections exactly as on disk -> arena on the device (CRC32C + BE decode in k_decode_sections):
    probes equal those over the host-decoded arena; a corrupt section is isolated per block with
    parseFilterSection's own error code and never poisons the others."""
    import struct
    rng = np.random.default_rng(91)
    n_blocks = 90
    plan, blocks_str, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.05, max_tokens=2000)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    sections = []
    for b in range(n_blocks):
        fl = []
        for c in range(3):
            d = plan.desc[b * 3 + c]
            if d["m"] == 0:
                fl.append(None)
            else:
                nw = O.words_for(int(d["m"]))
                fl.append(O.Filter(int(d["m"]), int(d["k"]), words[int(d["word_off"]): int(d["word_off"]) + nw]))
        sections.append(O.encode_filter_section(fl))
    cb = Q.compile_queries([None] + [H.random_expression(rng, vocab, None) for _ in range(400)])
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    aid, status = ctx.arena_load_sections(sections)
    assert not status.any()
    assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), want)
    ctx.arena_free(aid)

    # corruptions, one per block: flipped payload bit, flipped CRC byte, unknown flag bit (CRC fixed up),
    # truncated body (CRC fixed up), trailing bytes (CRC fixed up), too small, empty (= block without filters)
    def with_crc(payload):
        return payload + struct.pack("<I", O.crc32c(payload))
    bad = list(sections)
    s5 = bytearray(sections[5]); s5[len(s5) // 2] ^= 0x10; bad[5] = bytes(s5)
    s9 = bytearray(sections[9]); s9[-1] ^= 0xFF; bad[9] = bytes(s9)
    bad[12] = with_crc(bytes([sections[12][0] | 0x40]) + sections[12][1:-4])
    bad[20] = with_crc(sections[20][: len(sections[20]) // 2])
    bad[33] = with_crc(sections[33][:-4] + b"\x00\x00\x00")
    bad[40] = b"\x01\x02\x03"
    bad[41] = b""
    # crafted headers under a correct checksum: an m whose (m + 63) / 64 wraps, a bitset shorter than m, k = 0, k beyond the cap
    crafted = lambda m, k, blen: with_crc(bytes([1]) + struct.pack("<I", 32) + struct.pack(">QQQ", m, k, blen) + b"\xAA" * 8)
    bad[50], bad[51], bad[52], bad[53], bad[54] = (crafted(2 ** 64 - 1, 3, 64), crafted(2 ** 64 - 40, 3, 64), crafted(65, 3, 64),
                                                   crafted(64, 0, 64), crafted(64, 1025, 64))
    aid, status = ctx.arena_load_sections(bad)
    expect = {5: -2, 9: -2, 12: -3, 40: -1, 41: 0, 33: -6, 50: -5, 51: -5, 52: -5, 53: -5, 54: -5}
    for b, code in expect.items():
        assert status[b] == code, (b, status[b])
    assert status[20] in (-4, -5)
    for b in range(n_blocks):          # oracle parse agrees on ok / not ok for every block ...
        if 50 <= b <= 54:              # ... but these: bloom/v3 ReadFrom (the oracle) takes m, k and the bitset length as they come;
            continue                   # the device and the host codec call such a filter bad (deviations)
        try:
            O.parse_filter_section(bad[b]) if bad[b] else None
            ok = True
        except ValueError:
            ok = False
        assert ok == (status[b] == 0), b
    got = ctx.probe(aid, n_blocks, terms, ops, poff)
    ctx.arena_free(aid)
    failed = {b for b in range(n_blocks) if status[b] != 0} | {41}
    desc2 = plan.desc.copy()
    for b in failed:
        desc2["m"][b * 3: b * 3 + 3] = 0          # failed / missing sections behave as nil filters (fail-open)
    want2 = O.probe_batch(words, desc2.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(got, want2)
    # untouched blocks are bit-identical to the clean run
    mask = np.zeros(n_blocks, dtype=bool); mask[list(failed)] = True
    bits = lambda a: np.unpackbits(a.view(np.uint8), axis=1, bitorder="little")[:, :n_blocks]
    assert np.array_equal(bits(got)[:, ~mask], bits(want)[:, ~mask])
