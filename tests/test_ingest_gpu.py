"""Device ingest (bsg_ingest_*: rows -> distinct entries -> exact counts -> bitsets) vs the oracle.

The checker is the pure-Python walker oracle (oracle/walker_oracle.py: indexRow) for the entry sets and
the C oracle (oracle/bloom_oracle.c) for the bitsets; the product path is k_ingest_rows /
k_ingest_union / k_build_sets through the C-ABI, with the C++ host walker finishing the rows the
device walker hands back.  Row tables follow tokenizer_test.go:86-190 and
no_false_negatives_test.go:103-321 (same rows as tests/test_host_tables.py).
"""
import json

import numpy as np
import pytest

from bloomsearch_amd import host as Hst, ingest as I, query as Q, synth
from oracle import oracle as O
from oracle import walker_oracle as W
from tests.test_host_tables import JSON_MATCHING, KEYS, _random_value, go_marshal
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu

FPR = 0.001
TRUSTED = 1   # BSG_INGEST_TRUSTED_JSON


@pytest.fixture(params=[0, TRUSTED], ids=["validated", "trusted"])
def flags(request):
    """Valid-JSON scenarios run with and without the device's validation pass."""
    return request.param


def oracle_sets(rows):
    sets = (set(), set(), set())
    for r in rows:
        W.index_row(r, sets)
    return sets


def check_against_sets(res, set_index, sets, what=""):
    """counts exact and bitsets bit-identical to the oracle's build of the same entry sets"""
    for kind in range(3):
        assert int(res.counts[set_index, kind]) == len(sets[kind]), (what, kind, sorted(sets[kind])[:8])
        want = O.build_sized(sorted(sets[kind]), FPR)
        d = res.desc[set_index * 3 + kind]
        assert (int(d["m"]), int(d["k"])) == (want.m, want.k)
        assert np.array_equal(res.filter_words(set_index, kind), want.words), (what, kind)


def test_synthetic_log_rows_three_blocks_and_file(ctx, flags):
    row_sets = [synth.rows_json(b * 700, 700) for b in range(3)]
    res = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0, 0, 0], n_parents=1, flags=flags)
    assert len(res.fallback_rows) == 0            # printable ASCII, no escapes: all on the device
    assert res.stats.n_rows == 2100 and res.stats.ms_walk > 0
    union = (set(), set(), set())
    for b, rows in enumerate(row_sets):
        sets = oracle_sets(rows)
        check_against_sets(res, b, sets, "block %d" % b)
        for u, s in zip(union, sets):
            u |= s
    check_against_sets(res, 3, union, "file")     # parent = unionInto of the three blocks (flush.go:221,253)
    assert not res.status.any()


ROWS_DEVICE = [  # inside the device walker's envelope
    b'{"a":1}', b'{"a":{"b":{"c":"Deep VALUE here"}}}', b'{"a.b.c":"x","d":[1,2,[3,{"e":"f"}]]}',
    b'{".a":"xyz","id":4}', b'{"trail.":"v","x..y":"w","":"empty key"}', b'{"":{"":{"":"n"}}}',
    b'{"n":null,"t":true,"f":false,"z":0,"neg":-0.5e+3,"E":1E5,"big":9007199254740993}',
    b'{ "spaced" : [ "  two   words  " , "" , " " ] , "o" : { } , "arr" : [ ] }',
    b'{"a::b":"x::y","a":"b::c"}', b'{"dup":"A a A","dup":"b"}', b'[{"top":"array"},5,"str"]', b'"just a string"', b'12',
    b'{"MiXeD":"CamelCase UPPER lower 123ABC"}', b'{"k":"!@#$%^&*() ~`[]{};:,./<>?"}',
    b'{"user":{"name":"John Doe","tags":[{"type":"admin"},{"role":"user"}]}}',
    # JSON escapes that decode below 0x80 are the device's too (json.Marshal writes <, >, & as \\u003c, \\u003e, \\u0026)
    b'{"m":"\\u003chtml\\u003e\\u0026amp; x"}', b'{"back\\\\slash":"v","q?x":"y"}', b'{"t":"tab\\there","n":"new\\nline \\r\\f\\b x"}',
    b'{"q":"say \\"Hi\\" ok","s":"a\\/b"}', b'{"u":"\\u0041BC \\u0061\\u0020\\u0062 \\u003C\\u003c","z":"nul\\u0000in"}',
    b'{"a\\u002eb":"dot in key","k\\"q":1,"sp\\u0020ace":[true]}', b'{"e":"\\\\","f":"\\\\\\"x"}',
]
ROWS_UTF8_DEVICE = [   # valid UTF-8: Unicode white space, Unicode lower-casing, non-ASCII keys and \\uXXXX are the device's too
    '{"héllo":"日本語 ÀB"}'.encode(), '{"k":"über Ωmega ǅ İstanbul K"}'.encode(), '{"k":"Привет МИР"}'.encode(), b'{"e":"caf\\u00e9 \\u00c0 \\u212a"}', b'{"emoji":"\\ud83d\\ude00 x\\ud83c\\udf89y \\uD83D\\uDE80"}', b'{"\\ud83d\\ude00k":"surrogate pair in a key"}',
    '{"héllo":"日本語 café ñ 😀 ß straße"}'.encode(), '{"nbsp":"a\u00a0b\u2003c\u3000d\u0085e"}'.encode(), b'{"nbsp":"a\\u00a0b \\u00e9t\\u00e9 \\u4e2d\\u2028x"}',
    '{"ключ":"значение и ещё","مفتاح":"قيمة","k\\u00e9y":"v"}'.encode(), '{"mixed":"ASCII Upper ünï 中文 END"}'.encode(),
]
ROWS_HOST = [    # must be handed to the host walker
    b'{"raw":"ctl\x01char"}', b'{"a":\t1}', b'{"lone":"\\ud800 x"}', b'{"lone":"low first \\ude00\\ud83d"}', b'{"lone":"\\ud83d\\u0041"}',

    ('{' + '"a":{' * 17 + '"x":1' + '}' * 17 + '}').encode(),
    ('{"' + 'k' * 150 + '":{"' + 'j' * 60 + '":1}}').encode(),
    b'{"s":"\xff\xfe bad utf8"}', b'{"s":"overlong \xc0\xaf"}', b'{"s":"surrogate \xed\xa0\x80"}', b'{"s":"too big \xf5\x80\x80\x80"}',
    b'{"s":"cut \xe6\x97"}', b'{"s":"stray \x80 cont"}',
]
ROWS_MALFORMED = [b'{"a": [1, 2', b'{"a":"x" "b":1}', b'{"a":tru}', b'{"ok":"first","b":01x}', b'{"a":1}}', b'{"a":"unterminated',
                  b'', b'   ', b'{"a":1,}', b'{"k" 1}', b'{"a":-}', b'{"a":1.}', b'{"a":1e}', b'{"x":"y"} trailing',
                  b'{"a":"bad \\x escape"}', b'{"a":"short \\u12"}', b'{"a":"\\u00zz"}', b'{"k\\q":1}', b'{"a":"ok","b":"dangling\\']


def host_sets(rows):
    s = Hst.EntrySets()
    for r in rows:
        try:
            s.index_row(r)
        except Hst.HostError:
            pass
    return s.as_python_sets()


def test_row_tables_one_set_per_row(ctx, flags):
    dev = ROWS_DEVICE + ROWS_UTF8_DEVICE
    rows = dev + ROWS_HOST + [r.encode() for r, _ in JSON_MATCHING]
    res = I.device_ingest(ctx, [[r] for r in rows], FPR, flags=flags)
    fb = set(int(x) for x in res.fallback_rows)
    assert fb.isdisjoint(range(len(dev))), [rows[i] for i in sorted(fb) if i < len(dev)]
    assert set(range(len(dev), len(dev) + len(ROWS_HOST))) <= fb, [rows[i] for i in range(len(dev), len(dev) + len(ROWS_HOST)) if i not in fb]
    for i, r in enumerate(rows):
        want = host_sets([r])
        try:
            r.decode("utf-8")                     # the Python oracle only speaks strict UTF-8; and it keeps a lone \\ud800 escape
            if b'"lone"' not in r:                # where Go (and the host walker) write U+FFFD
                assert want == W.index_row(r), r  # host walker == Python oracle wherever the oracle parses the row
        except (ValueError, UnicodeDecodeError):
            pass
        check_against_sets(res, i, want, r)


def test_malformed_rows_keep_what_the_host_walker_keeps(ctx):
    # a flagged row may have inserted a prefix of its entries on the device; the host walker re-inserts its own
    # (lenient) prefix and the union must equal the host-only result
    res = I.device_ingest(ctx, [[r] for r in ROWS_MALFORMED], FPR)
    fb = set(int(x) for x in res.fallback_rows)
    for i, r in enumerate(ROWS_MALFORMED):
        check_against_sets(res, i, host_sets([r]), r)
    assert fb == set(range(len(ROWS_MALFORMED)))


def test_random_rows_mixed_device_and_host(ctx, flags):
    # the property generator of no_false_negatives_test.go:398-459 (re-seeded): ~half the rows carry escapes / UTF-8
    rng = np.random.default_rng(11)
    row_sets, plain = [], 0
    for s in range(12):
        rows = []
        for _ in range(40):
            obj = {KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}
            rows.append(go_marshal(obj))
        row_sets.append(rows)
    parents = [s % 2 for s in range(12)]
    res = I.device_ingest(ctx, row_sets, FPR, parent_of_set=parents, n_parents=2, flags=flags)
    n_rows = sum(len(r) for r in row_sets)
    assert len(res.fallback_rows) < n_rows
    unions = [(set(), set(), set()), (set(), set(), set())]
    for s, rows in enumerate(row_sets):
        sets = oracle_sets(rows)
        check_against_sets(res, s, sets, "set %d" % s)
        for u, x in zip(unions[parents[s]], sets):
            u |= x
    check_against_sets(res, 12, unions[0], "parent 0")
    check_against_sets(res, 13, unions[1], "parent 1")


def test_ascii_fuzz_all_on_device(ctx, flags):
    # random printable-ASCII documents with random spacing: nothing may fall back, everything must match
    rng = np.random.default_rng(5)
    alphabet = [chr(c) for c in range(0x20, 0x7F) if chr(c) not in '"\\']

    def rand_text(n):
        return "".join(alphabet[rng.integers(0, len(alphabet))] for _ in range(n))

    def rand_value(depth):
        r = rng.random()
        if depth < 5 and r < 0.25:
            return {rand_text(rng.integers(0, 9)): rand_value(depth + 1) for _ in range(rng.integers(0, 4))}
        if depth < 5 and r < 0.4:
            return [rand_value(depth + 1) for _ in range(rng.integers(0, 4))]
        if r < 0.55:
            return int(rng.integers(-10 ** 12, 10 ** 12))
        if r < 0.6:
            return [None, True, False][rng.integers(0, 3)]
        return rand_text(rng.integers(0, 40))

    rows = []
    for _ in range(600):
        obj = {rand_text(rng.integers(0, 12)): rand_value(0) for _ in range(rng.integers(0, 6))}
        sep = [(",", ":"), (", ", ": "), (" , ", " : ")][rng.integers(0, 3)]
        rows.append(json.dumps(obj, separators=sep).encode())
    row_sets = [rows[i::6] for i in range(6)]
    res = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0] * 6, n_parents=1, flags=flags)
    assert len(res.fallback_rows) == 0
    union = (set(), set(), set())
    for s, rs in enumerate(row_sets):
        sets = oracle_sets(rs)
        check_against_sets(res, s, sets, "set %d" % s)
        for u, x in zip(union, sets):
            u |= x
    check_against_sets(res, 6, union, "file")


def test_unicode_fuzz_device_and_host_agree_with_the_oracle(ctx, flags):
    # random text over several scripts, Unicode white space, cased runes (=> host) and emoji; keys too
    rng = np.random.default_rng(17)
    pools = ["abcXYZ019-_.", "éñüßøåçœ", "ÀÉÜÑØÅ", "日本語中文한국어", "абвгд", "АБВГД", "αβγδ", "ΑΒΓΔ", "😀🎉🚀", " \u00a0\u2003\u3000\u0085\t",
             "\u1680\u2028\u2029\u202f\u205f", "ǅǈǋ", "İıſK"]

    def rand_text(n):
        out = []
        for _ in range(n):
            p = pools[rng.integers(0, len(pools))]
            out.append(p[rng.integers(0, len(p))])
        return "".join(out)

    row_sets = []
    for s_ in range(8):
        rows = []
        for _ in range(120):
            obj = {rand_text(rng.integers(1, 6)): (rand_text(rng.integers(0, 24)) if rng.random() < 0.8 else [rand_text(3), int(rng.integers(0, 99))])
                   for _ in range(rng.integers(1, 5))}
            rows.append(json.dumps(obj, ensure_ascii=bool(rng.random() < 0.3), separators=(",", ":")).encode())
        row_sets.append(rows)
    res = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0] * 8, n_parents=1, flags=flags)
    n_rows = sum(len(r) for r in row_sets)
    assert len(res.fallback_rows) == 0                  # valid UTF-8 and well-formed escapes: all on the device
    union = (set(), set(), set())
    for s_, rows in enumerate(row_sets):
        sets = oracle_sets(rows)
        assert host_sets(rows) == sets
        check_against_sets(res, s_, sets, "set %d" % s_)
        for u, x in zip(union, sets):
            u |= x
    check_against_sets(res, 8, union, "file")


def test_unicode_14_case_pairs_fold_on_the_device_and_unknown_code_points_go_to_the_host(ctx, flags):
    """Rows with cased letters Unicode 14.0 added fold on the device exactly as the oracle's list says (what Go >= 1.21
    does); a code point the tables do not know (unassigned in the Unicode they were generated from: here U+A7CB and
    U+10D50, both cased letters since Unicode 16.0) sends the row to the host, whose own unicode.ToLower decides."""
    ups = ["\u2c2f", "\ua7c0", "\ua7d0", "\ua7d6", "\ua7d8", "\U00010570", "\U0001057c", "\U0001058c", "\U00010595"]
    rows = [json.dumps({"msg": "Tok%s %s END" % (u, u), "k%s" % u: "v"}, ensure_ascii=bool(i & 1), separators=(",", ":")).encode()
            for i, u in enumerate(ups)]
    res = I.device_ingest(ctx, [rows], FPR, flags=flags)
    assert len(res.fallback_rows) == 0
    sets = oracle_sets(rows)
    assert "tok\u2c5f" in sets[1] and "\U00010597" in sets[1]
    check_against_sets(res, 0, sets, "unicode 14")
    unknown = [json.dumps({"msg": "a%sb" % u}, ensure_ascii=bool(i & 1), separators=(",", ":")).encode()
               for i, u in enumerate(["\ua7cb", "\U00010d50", "\ua7cb"])]
    res = I.device_ingest(ctx, [rows[:2] + unknown], FPR, flags=flags)
    assert list(res.fallback_rows) == [2, 3, 4]
    # the host walker of this mirror finishes them (identity fold): nothing is lost
    assert int(res.counts[0, 1]) == len(oracle_sets(rows[:2] + unknown)[1])
    hits, handed_back = ctx.match_rows(unknown + rows[:1], Q.CompiledMatcher(Q.Token("tok\u2c5f")))
    assert list(handed_back) == [0, 1, 2] and list(hits) == [False, False, False, True]


def test_chunked_upload_overlapping_the_walk(ctx, flags):
    """bsg_ingest_rows uploads the rows in chunks and walks chunk i while chunk i+1 is in flight: with a 64 KiB chunk
    a few thousand rows become dozens of chunks — sets spanning chunk boundaries, tables that must grow in the middle
    (tiny hint => re-runs of single chunks), rows for the host in several chunks.  Counts, bitsets and the fallback list
    must be exactly those of the one-chunk run."""
    rng = np.random.default_rng(99)
    row_sets = []
    for s_ in range(5):
        rows = synth.rows_json(s_ * 700, 700)
        for j in range(0, 700, 97):                       # rows the device hands back (nested deeper than it follows), spread over the chunks
            rows[j] = b'{"d":' * 20 + b'"deep%d"' % j + b"}" * 20
        row_sets.append(rows)
    try:
        ctx.set_ingest_chunk(1 << 16)
        small = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0] * 5, n_parents=1, flags=flags, slots_hint=[64] * 15)
    finally:
        ctx.set_ingest_chunk(0)
    whole = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0] * 5, n_parents=1, flags=flags)
    assert small.stats.table_grows > 0
    assert list(small.fallback_rows) == list(whole.fallback_rows) and len(whole.fallback_rows) == 5 * 8
    assert np.array_equal(small.counts, whole.counts)
    for s_, rows in enumerate(row_sets):
        check_against_sets(small, s_, oracle_sets(rows), "set %d" % s_)


def test_tables_grow_from_a_tiny_hint(ctx, flags):
    rows = synth.rows_json(0, 1500)
    res = I.device_ingest(ctx, [rows], FPR, parent_of_set=[0], n_parents=1, slots_hint=[64, 64, 64], flags=flags)
    assert res.stats.table_grows >= 2
    sets = oracle_sets(rows)
    check_against_sets(res, 0, sets, "grown")
    check_against_sets(res, 1, sets, "parent of one")


@pytest.mark.parametrize("binned", [True, False], ids=["binned", "global-atomics"])
def test_large_sets_lds_staged_above_64k_and_beyond_lds(ctx, binned):
    # 60 000 distinct tokens -> a 108 KB bitset (LDS-staged, needs the opt-in above 64 KB); 130 000 -> 234 KB: beyond LDS,
    # assembled window by window from binned locations (k_bin_*), or — the path m >= 2^31 still takes — with global
    # atomicOr, sliced over several workgroups; the parent holds their union
    sets = [[b'{"id":"u%d"}' % i for i in range(60000)], [b'{"id":"v%d","n":%d}' % (i, i % 7) for i in range(130000)]]
    ctx.set_lab(2, (16 << 30) if binned else 0)
    ctx.set_lab(6, 0)                                          # (by default only bitsets with millions of locations are binned)
    try:
        res = I.device_ingest(ctx, sets, FPR, parent_of_set=[0, 0], n_parents=1, flags=TRUSTED)
    finally:
        ctx.set_lab(2, 16 << 30)
        ctx.set_lab(6, 4 << 20)
    assert len(res.fallback_rows) == 0
    want = [({"id"}, {"u%d" % i for i in range(60000)}, {"id::u%d" % i for i in range(60000)}),
            ({"id", "n"}, {"v%d" % i for i in range(130000)} | {str(i) for i in range(7)},
             {"id::v%d" % i for i in range(130000)} | {"n::%d" % i for i in range(7)})]
    for i, w in enumerate(want):
        check_against_sets(res, i, w, "large %d" % i)
    check_against_sets(res, 2, tuple(a | b for a, b in zip(*want)), "parent")


def test_resident_arenas_equal_reloaded_sections(ctx):
    """bsg_ingest_build_sections leaves the filters it just wrote resident as probe arenas: probing them must give the
    same survivors as uploading + decoding the returned section bytes, and the oracle's answer; the sections themselves
    must be the oracle's bytes."""
    from bloomsearch_amd import query as Q
    from tests import helpers as H
    row_sets = [synth.rows_json(b * 400, 400) for b in range(5)] + [[]]
    first = np.zeros(len(row_sets) + 1, dtype=np.uint32)
    first[1:] = np.cumsum([len(r) for r in row_sets])
    ing = ctx.ingest_rows([r for rs in row_sets for r in rs], first, [0, 0, 0, 1, 1, 1], 2, flags=TRUSTED)
    counts, _ = ctx.ingest_finish(ing, 8)
    desc, _ = I.plan_desc(counts, FPR)
    secs, a_sets, a_parents = ctx.ingest_build_sections(ing, desc, arenas=True)
    ctx.ingest_free(ing)
    assert len(secs) == 8
    sets = [oracle_sets(rs) for rs in row_sets]
    sets += [tuple(set().union(*(sets[i][k] for i in grp)) for k in range(3)) for grp in ((0, 1, 2), (3, 4, 5))]
    for i, ss in enumerate(sets):
        assert secs[i] == O.encode_filter_section([O.build_sized(sorted(ss[k]), FPR) for k in range(3)]), i
    d = synth.draws(0, 40)
    exprs = [Q.And(Q.FieldToken("level", synth.LEVELS[d["level"][i]]), Q.FieldToken("user_id", str(int(d["user_id"][i]))))
             for i in range(40)] + [Q.Token("absent-token"), Q.Field("nested.az"), None]
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    bid = ctx.batch_create(terms, ops, poff)
    for arena, lo, n in ((a_sets, 0, 6), (a_parents, 6, 2)):
        got = ctx.probe_batch(arena, bid, cb.n_queries, n)
        reloaded, status = ctx.arena_load_sections(secs[lo: lo + n])
        assert not status.any()
        assert np.array_equal(got, ctx.probe_batch(reloaded, bid, cb.n_queries, n))
        ctx.arena_free(reloaded)
        for q, e in enumerate(exprs):               # oracle: evaluate the expression over the sets' filters
            for b in range(n):
                fl = [O.build_sized(sorted(sets[lo + b][k]), FPR) for k in range(3)]
                def ev(x):
                    if x is None:
                        return True
                    if x["ExpressionType"] == "CONDITION":
                        kind, sv = Q.term_of(x["Condition"])
                        return fl[kind].test(sv)
                    vals = [ev(c) for c in x["Children"]]
                    return all(vals) if x["ExpressionType"] == "AND" else any(vals)
                assert bool((int(got[q, b >> 6]) >> (b & 63)) & 1) == ev(e), (q, b)
        ctx.arena_free(arena)
    ctx.batch_free(bid)


@pytest.mark.parametrize("n_entries", [2, 3])
def test_resident_arenas_on_a_sharded_context(n_entries):
    """The same on a context over several entries: the filters are built on the first one and every entry receives its
    blocks (b % n) device to device; probing the resident arenas must equal probing the reloaded sections."""
    from bloomsearch_amd import query as Q
    from bloomsearch_amd.gpu import Context
    from tests import helpers as H
    with Context(device_ids(n_entries)) as mctx:
        row_sets = [synth.rows_json(b * 150, 150) for b in range(11)] + [[]]
        first = np.zeros(len(row_sets) + 1, dtype=np.uint32)
        first[1:] = np.cumsum([len(r) for r in row_sets])
        ing = mctx.ingest_rows([r for rs in row_sets for r in rs], first, [i % 3 for i in range(12)], 3, flags=TRUSTED)
        counts, _ = mctx.ingest_finish(ing, 15)
        desc, _ = I.plan_desc(counts, FPR)
        secs, a_sets, a_parents = mctx.ingest_build_sections(ing, desc, arenas=True)
        mctx.ingest_free(ing)
        assert a_sets != 0 and a_parents != 0
        d = synth.draws(0, 30)
        exprs = [Q.And(Q.FieldToken("level", synth.LEVELS[d["level"][i]]), Q.FieldToken("user_id", str(int(d["user_id"][i])))) for i in range(30)]
        exprs += [Q.Token("absent-token"), Q.Field("nested.az"), None, Q.FieldToken("service", "billing")]
        for batch in (exprs, exprs[:1]):                      # many terms (streaming kernels) and one query (one dispatch per entry)
            cb = Q.compile_queries(batch)
            ops, poff, _ = cb.arrays()
            terms = H.gpu_terms(mctx, cb)
            bid = mctx.batch_create(terms, ops, poff)
            for arena, lo, n in ((a_sets, 0, 12), (a_parents, 12, 3)):
                reloaded, status = mctx.arena_load_sections(secs[lo: lo + n])
                assert not status.any()
                assert np.array_equal(mctx.probe_batch(arena, bid, cb.n_queries, n), mctx.probe_batch(reloaded, bid, cb.n_queries, n))
                mctx.arena_free(reloaded)
            mctx.batch_free(bid)
        mctx.arena_free(a_sets)
        mctx.arena_free(a_parents)


def test_rows_from_pinned_host_memory(ctx):
    rows = synth.rows_json(0, 800)
    blob = np.frombuffer(b"".join(rows), dtype=np.uint8)
    off = np.zeros(len(rows) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in rows])
    pinned = ctx.pinned_array(len(blob))
    pinned[:] = blob
    ing = ctx.ingest_rows((pinned, off), [0, 800], flags=TRUSTED)
    counts, status = ctx.ingest_finish(ing, 1)
    desc, n_words = I.plan_desc(counts, FPR)
    words = ctx.ingest_build(ing, desc, n_words)
    ctx.ingest_free(ing)
    ctx.pinned_free(pinned)
    sets = oracle_sets(rows)
    res = I.IngestResult(counts, status, desc, words, None, np.zeros(0, dtype=np.uint32))
    check_against_sets(res, 0, sets, "pinned")


def test_empty_sets_and_rowless_ingest(ctx):
    res = I.device_ingest(ctx, [[], [b'{"a":"b"}'], []], FPR, parent_of_set=[0, 0, 1], n_parents=2)
    empty = (set(), set(), set())
    check_against_sets(res, 0, empty)             # n' = max(0, 1): m = 15 bits, none set (ingest.go:135-140)
    check_against_sets(res, 1, oracle_sets([b'{"a":"b"}']))
    check_against_sets(res, 2, empty)
    check_against_sets(res, 3, oracle_sets([b'{"a":"b"}']))
    check_against_sets(res, 4, empty)


def test_ingest_argument_errors(ctx):
    from bloomsearch_amd._lib import BloomGpuError
    with pytest.raises(BloomGpuError):
        ctx.ingest_rows([b'{}'], [0, 2])                      # set_first_row does not span the rows
    with pytest.raises(BloomGpuError):
        ctx.ingest_rows([b'{}'], [0, 1], parent_of_set=[3], n_parents=1)
    with pytest.raises(BloomGpuError):
        ctx.ingest_finish(123456, 1)
    ing = ctx.ingest_rows([b'{"a":1}'], [0, 1])
    with pytest.raises(BloomGpuError):
        ctx.ingest_build(ing, np.zeros(3, dtype=I.DESC_DTYPE), 2)   # finish has not run
    ctx.ingest_free(ing)
    with pytest.raises(BloomGpuError):
        ctx.ingest_free(ing)


def test_every_chunk_alignment(ctx):
    """The walker reads rows in 8-byte chunks wherever they start: every tricky row is placed at all eight alignments
    (a filler row of varying length in front), alone in its set, and must give the host walker's entry sets each time —
    escapes, UTF-8 sequences, numbers and literals cut by chunk boundaries at every possible byte."""
    tricky = ROWS_DEVICE + ROWS_UTF8_DEVICE + [r.encode() for r, _ in JSON_MATCHING[:12]]
    rows, sets_of = [], []
    for t in tricky:
        for pad in range(8):
            rows.append(b'{"f":"' + b"x" * (pad + 1) + b'"}')
            rows.append(t)
    res = I.device_ingest(ctx, [[r] for r in rows], FPR, flags=TRUSTED)
    assert len(res.fallback_rows) == 0
    cache = {}
    for i, r in enumerate(rows):
        if r not in cache:
            cache[r] = host_sets([r])
        want = cache[r]
        for kind in range(3):
            assert int(res.counts[i, kind]) == len(want[kind]), (r, kind)
    # bitsets of one alignment family in full
    for i in range(1, 16, 2):
        check_against_sets(res, i, cache[rows[i]], rows[i])


def test_file_level_union_in_lds_partitions_equals_global_tables_retry_and_fallback(ctx):
    """flush.go:221,253: the file's entry sets are the unions of its blocks'.  k_union_partitions deduplicates them partition by
    partition in LDS (dense parents, one atomic per workgroup); lab key 9 = 1 takes round 2's global hash tables, key 10 starts
    the partitioning 2^v x too coarse so that the 4-x-finer retry (v = 4) and the fall-back to the global tables (v = 14) run.
    Counts, statuses and bitsets must be identical on every route — and equal the oracle's build of the union."""
    row_sets = [synth.rows_json(b * 900, 900) for b in range(64)] + [[]]
    parents = [0] * 52 + [1] * 12 + [1]          # file 0: ~95 k distinct tokens — more than 64 LDS partitions hold, so v = 14 cannot be rescued by three 4-x-finer retries
    routes = {}
    try:
        # 32 + v: 2^v x FINER than needed: a partition's run in a child shrinks to 2 slots, 1 slot, a fraction of a slot — the probe
        # spill behind a run then crosses several partitions (a 2-slot run at load 0.6 miscounted partition 0 before its fix)
        for name, mode, coarsen in (("partitions", 0, 0), ("global tables", 1, 0), ("retry", 0, 4), ("fallback", 0, 14), ("finer x16", 0, 36),
                                    ("finer x32", 0, 37), ("finer x128", 0, 39), ("finer x1024", 0, 42)):
            ctx.set_lab(9, mode)
            ctx.set_lab(10, coarsen)
            routes[name] = I.device_ingest(ctx, row_sets, FPR, parent_of_set=parents, n_parents=3)      # parent 2 has no children at all
    finally:
        ctx.set_lab(9, 0)
        ctx.set_lab(10, 0)
    base = routes["partitions"]
    assert not base.status.any() and base.stats.ms_union > 0
    assert routes["retry"].stats.table_grows >= 1 and routes["fallback"].stats.table_grows >= 4, (routes["retry"].stats.table_grows, routes["fallback"].stats.table_grows)
    for name, res in routes.items():
        assert np.array_equal(res.counts, base.counts), name
        assert np.array_equal(res.status, base.status), name
        assert np.array_equal(res.words, base.words), name
    assert base.stats.table_bytes < routes["global tables"].stats.table_bytes       # 40 B per child entry vs 2 x rounded up to a power of two
    unions = [(set(), set(), set()) for _ in range(3)]
    for s, rows in enumerate(row_sets):
        for u, x in zip(unions[parents[s]], oracle_sets(rows)):
            u |= x
    for p in range(3):
        check_against_sets(base, len(row_sets) + p, unions[p], "file %d" % p)
    assert [int(x) for x in base.counts[len(row_sets) + 2]] == [0, 0, 0]
