"""The final row test on the device (bsg_match_rows: compileRowMatcher / matchRowBytes, row_matcher.go:257-626) vs the
oracle's set-based matcher (oracle/walker_oracle.py matches_bloom_expression) and the C++ host matcher.

Tables: tokenizer_test.go:86-190 (TestJSONMatching), query_test.go:91-111, no_false_negatives_test.go:103-321,
row_matcher_test.go:38-41,99-100 — the same rows and verdicts as tests/test_host_tables.py, through the GPU.
"""
import json

import numpy as np
import pytest

from bloomsearch_amd import host as Hst, query as Q, synth
from oracle import walker_oracle as W
from tests import helpers as H
from tests.test_host_tables import JSON_MATCHING, KEYS, WORDS, _expr, _random_value, go_marshal
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu


def device_match(ctx, rows, expr):
    """device verdicts, the host matcher deciding the rows the device hands back"""
    got, fb = ctx.match_rows(rows, Q.CompiledMatcher(expr))
    for r in fb:
        assert not got[r]
        got[r] = Hst.match_row(expr, rows[int(r)])
    return got, fb


def test_json_matching_tables_on_device(ctx):
    for row, cases in JSON_MATCHING:
        raw = row.encode()
        for kind, args, want in cases:
            e = _expr(kind, args)
            got, _ = device_match(ctx, [raw], e)
            assert bool(got[0]) == want, (row, kind, args)
            assert Hst.match_row(e, raw) == want and W.matches_bloom_expression(raw, e) == want


def test_field_token_is_a_pair_not_the_joined_key(ctx):
    # row_matcher.go:587 / doc :296-301: FieldToken("a", "b::c") must not match {"a::b": "c"} although the bloom keys coincide
    rows = [b'{"a::b":"c"}', b'{"a":"b::c"}', b'{"a":"x","b":"y"}', b'{"a":["p","q"],"b":{"a":"q2"}}']
    cases = [(Q.FieldToken("a", "b::c"), [False, True, False, False]), (Q.FieldToken("a::b", "c"), [True, False, False, False]),
             (Q.FieldToken("a", "y"), [False, False, False, False]),       # token under another field of the same row
             (Q.FieldToken("a", "q"), [False, False, False, True]), (Q.FieldToken("b.a", "q2"), [False, False, False, True]),
             (Q.And(Q.FieldToken("a", "x"), Q.FieldToken("b", "y")), [False, False, True, False])]
    for e, want in cases:
        got, fb = device_match(ctx, rows, e)
        assert len(fb) == 0 and list(map(bool, got)) == want, e
        assert [Hst.match_row(e, r) for r in rows] == want     # (the oracle's set-based helper joins the key: tokenizer.go:236-330)


def test_targets_are_never_normalised_and_rows_are_folded(ctx):
    rows = [b'{"name":"ALICE Smith"}', '{"name":"Ünï ÀB ΩMEGA"}'.encode(), b'{"n":1E5,"t":true,"z":null}']
    for e, want in [(Q.Token("alice"), [True, False, False]), (Q.Token("ALICE"), [False, False, False]),
                    (Q.Token("ünï"), [False, True, False]), (Q.Token("àb"), [False, True, False]), (Q.Token("ωmega"), [False, True, False]),
                    (Q.FieldToken("n", "1e5"), [False, False, True]), (Q.FieldToken("n", "100000"), [False, False, False]),
                    (Q.FieldToken("t", "true"), [False, False, True]), (Q.Field("z"), [False, False, True]),
                    (Q.FieldToken("z", "null"), [False, False, False])]:
        got, fb = device_match(ctx, rows, e)
        assert len(fb) == 0 and list(map(bool, got)) == want, e


def test_expression_semantics(ctx):
    # evalMatcherNode (row_matcher.go:257-290): nil => true, And() => true, Or() => false, unknown => false, nil condition => true
    rows = [b'{"a":"x"}', b'{"b":"y"}', b'{}', b'[1,2]']
    unknown_expr = {"ExpressionType": "XOR", "Children": []}
    unknown_cond = {"ExpressionType": "CONDITION", "Condition": {"Type": "BOGUS", "Field": "a", "Token": "x"}}
    nil_cond = {"ExpressionType": "CONDITION", "Condition": None}
    for e, want in [(None, [True] * 4), (Q.And(), [True] * 4), (Q.Or(), [False] * 4), (unknown_expr, [False] * 4),
                    (unknown_cond, [False] * 4), (nil_cond, [True] * 4), (Q.Or(unknown_cond, Q.Field("a")), [True, False, False, False]),
                    (Q.And(nil_cond, Q.Token("y")), [False, True, False, False]),
                    (Q.And(Q.Or(Q.Field("a"), Q.Field("b")), Q.Or(Q.Token("x"), Q.And())), [True, True, False, False])]:
        got, fb = device_match(ctx, rows, e)
        assert len(fb) == 0 and list(map(bool, got)) == want, e
        assert [Hst.match_row(e, r) for r in rows] == want


def test_random_rows_random_expressions_vs_oracle_and_host(ctx):
    rng = np.random.default_rng(41)
    rows = []
    for _ in range(700):
        obj = {KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}
        rows.append(go_marshal(obj))
    vocab = set()
    paths = set()
    for r in rows[:200]:
        f, t, _ = W.index_row(r)
        vocab |= t
        paths |= f
    vocab, paths = sorted(vocab), sorted(paths)

    def rand_expr(depth=0):
        r = rng.random()
        if depth >= 3 or r < 0.5:
            k = rng.integers(0, 3)
            tok = vocab[rng.integers(0, len(vocab))] if rng.random() < 0.85 else "absent%d" % rng.integers(0, 99)
            fld = paths[rng.integers(0, len(paths))] if rng.random() < 0.85 else "nope.%d" % rng.integers(0, 9)
            return [Q.Field(fld), Q.Token(tok), Q.FieldToken(fld, tok)][k]
        kids = [rand_expr(depth + 1) for _ in range(int(rng.integers(0, 4)))]
        return Q.And(*kids) if rng.random() < 0.5 else Q.Or(*kids)

    n_pos = 0
    for _ in range(40):
        e = rand_expr()
        got, fb = device_match(ctx, rows, e)
        assert len(fb) < len(rows) / 4
        want = [Hst.match_row(e, r) for r in rows]              # production semantics: (path, token) pairs
        assert list(map(bool, got)) == want, e
        if "::" not in json.dumps(e):                           # without "::" in a target the oracle's joined-key sets agree
            assert [W.matches_bloom_expression(r, e) for r in rows] == want
        n_pos += sum(want)
    assert n_pos > 100


def test_rows_the_device_hands_back(ctx):
    rows = [b'{"s":"\\xff\\xfe bad utf8 token"}', b'{"lone":"\\ud800 x"}', b'{"a":"ok"}', b'{"a": [1, 2', ('{' + '"a":{' * 17 + '"x":1' + '}' * 17 + '}').encode()]
    got, fb = ctx.match_rows(rows, Q.CompiledMatcher(Q.Token("ok")))
    assert sorted(int(x) for x in fb) == [0, 1, 3, 4] and list(map(bool, got)) == [False, False, True, False, False]


def test_block_of_log_rows_and_limits(ctx):
    from bloomsearch_amd._lib import BloomGpuError
    rows = synth.rows_json(0, 10000)
    d = synth.draws(0, 10000)
    e = Q.And(Q.FieldToken("level", "error"), Q.Or(Q.FieldToken("nested.region", "region-3"), Q.Token(str(int(d["user_id"][17])))))
    got, fb = ctx.match_rows(rows, Q.CompiledMatcher(e))
    assert len(fb) == 0 and ctx.last_match_ms() > 0
    want = (d["level"] == synth.LEVELS.index("error")) & ((d["region"] == 3) | (d["user_id"] == d["user_id"][17]) |
                                                        np.array([str(int(d["user_id"][17])) == str(synth.TS_BASE + i) for i in range(10000)]))
    assert np.array_equal(got, want) and want.sum() > 100
    with pytest.raises(BloomGpuError):
        ctx.match_rows(rows[:2], Q.CompiledMatcher(Q.And(*[Q.Token("t%d" % i) for i in range(65)])))   # > 64 conditions
    assert ctx.match_rows([], Q.CompiledMatcher(Q.Token("x")))[0].shape == (0,)


def test_compiled_matcher_equivalence_with_derived_queries(ctx):
    """row_matcher_test.go:31-172 (TestCompiledMatcherEquivalence): random rows plus the Unicode folding rows (U+212A,
    U+0130, NBSP), ~15 queries derived from each row's own entries (so most are positives) — the device matcher, the
    host matcher and (where no "::" is involved) the set-based oracle must agree on every (row, query)."""
    rng = np.random.default_rng(11)
    rows = [go_marshal({KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}) for _ in range(50)]
    rows += ['{"unit":"10 K","city":"İstanbul İZMİR","sp":"a b c"}'.encode(), '{"k":"Straße ÄÖÜ Ω"}'.encode()]
    queries = []
    for r in rows:
        fields, tokens, fts = (sorted(x) for x in W.index_row(r))
        qs = [Q.Field(f) for f in fields[:3]] + [Q.Token(t) for t in tokens[:3]]
        qs += [Q.FieldToken(*ft.split("::", 1)) for ft in fts[:3] if ft.count("::") == 1]
        if fields and tokens:
            qs += [Q.And(Q.Field(fields[0]), Q.Token(tokens[-1])), Q.Or(Q.Token("zz-absent"), Q.Field(fields[-1])),
                   Q.And(Q.Token(tokens[0]), Q.Or(Q.Field("nope"), Q.Token(tokens[-1]))), Q.Token(tokens[0].upper() + "X"),
                   Q.And(Q.Field(fields[0]), Q.Token("zz-absent")), Q.FieldToken(fields[-1], tokens[0])]
        queries.append(qs)
    # one device call per query over ALL rows: each query is a positive for its own row and mostly a negative elsewhere
    n_pos = 0
    for i, qs in enumerate(queries):
        for q in qs:
            got, fb = device_match(ctx, rows, q)
            want = [Hst.match_row(q, r) for r in rows]
            assert list(map(bool, got)) == want, (rows[i], q)
            if "::" not in json.dumps(q) and all(b"::" not in r for r in rows):
                assert [W.matches_bloom_expression(r, q) for r in rows] == want
            n_pos += sum(want)
    assert n_pos > 400


def test_chunked_upload_overlapping_the_match(ctx):
    """The rows travel in chunks on the copy stream while the chunk before is matched (bsg_match_rows, as bsg_ingest_rows
    does): verdicts, the rows handed back (indices of the CALL, whatever chunk they sat in) and the device time must not
    depend on where the chunks were cut — chunk sizes from a few rows to everything in one piece, on single- and
    multi-entry contexts (whose parts are chunked in turn)."""
    from bloomsearch_amd.gpu import Context
    rows = synth.rows_json(20000, 6000)
    bad = [b'{"s":"\\xff\\xfe bad utf8 token"}', b'{"a": [1, 2', ('{' + '"a":{' * 17 + '"x":1' + '}' * 17 + '}').encode()]
    where = [0, 255, 256, 257, 1023, 1024, 3000, 5998]
    for i, r in enumerate(where):
        rows[r] = bad[i % len(bad)]
    d = synth.draws(20000, 6000)
    e = Q.Or(Q.And(Q.FieldToken("level", "error"), Q.FieldToken("nested.region", "region-3")), Q.Token("ok"))
    want = (d["level"] == synth.LEVELS.index("error")) & (d["region"] == 3)
    want[where] = False
    got0, fb0 = ctx.match_rows(rows, Q.CompiledMatcher(e))
    assert sorted(int(x) for x in fb0) == where and np.array_equal(got0, want) and want.sum() > 50
    try:
        for chunk in (1 << 16, 70001, 1 << 18, 1 << 30):                # (the library takes no chunk below 64 KiB)
            ctx.set_ingest_chunk(chunk)
            got, fb = ctx.match_rows(rows, Q.CompiledMatcher(e))
            assert np.array_equal(got, got0) and sorted(int(x) for x in fb) == where, chunk
            assert ctx.last_match_ms() > 0
    finally:
        ctx.set_ingest_chunk(0)
    with Context(device_ids(3)) as m:
        m.set_lab(8, 1)                                     # shard whatever the size
        m.set_ingest_chunk(1 << 16)
        got, fb = m.match_rows(rows, Q.CompiledMatcher(e))
        assert np.array_equal(got, got0) and sorted(int(x) for x in fb) == where
