"""Host-side mirror (C++) vs the reference's language-neutral known-answer tables and vs the
pure-Python walker oracle.  CPU-only: nothing here touches the GPU.

Vectors transcribed (inputs + expected outputs only) from the reference's tests:
  tokenizer_test.go:10-84 (TestBasicWhitespaceTokenizer), :86-190 (TestJSONMatching),
  query_test.go:91-111 (bloom semantics over sample rows),
  no_false_negatives_test.go:103-321 (large ints, dotted keys, metachar keys, null, value encodings),
  row_matcher_test.go:38-41,99-100 (Unicode folding rows; targets are never lowercased).
"""
import json

import numpy as np
import pytest

from bloomsearch_amd import host as Hst, query as Q
from oracle import walker_oracle as W

TOKENIZER_TABLE = [
    ("hello world 123", ["hello", "world", "123"]),
    ("hello@world.com!test", ["hello@world.com!test"]),
    ("hello-world_test", ["hello-world_test"]),
    ("hello \U0001F60A world \U0001F389", ["hello", "\U0001F60A", "world", "\U0001F389"]),
    ("user@domain.com, password123!", ["user@domain.com,", "password123!"]),
    ("42", ["42"]),
    ("true", ["true"]),
    ("", []),
    ("hello   world", ["hello", "world"]),
    ("!@#$%^&*()", ["!@#$%^&*()"]),
    ("hello\tworld\ntest", ["hello", "world", "test"]),
    ("user-name_123@example.com (active)", ["user-name_123@example.com", "(active)"]),
]


@pytest.mark.parametrize("text,expected", TOKENIZER_TABLE)
def test_basic_whitespace_tokenizer_table(text, expected):
    assert Hst.tokenize(text) == expected
    assert W.basic_whitespace_lower_tokenizer(text) == expected


def test_unicode_14_case_pairs_fold_like_go_1_21_and_later():
    """Cased letters added in Unicode 14.0 (Python 3.10's tables stop at 13.0): one row per added range, host twin
    against the oracle's independently transcribed list (Go >= 1.21 ships Unicode 15.0; the reference needs Go 1.26)."""
    cases = [("\u2c2f", "\u2c5f"),                              # Glagolitic
             ("\ua7c0", "\ua7c1"), ("\ua7d0", "\ua7d1"), ("\ua7d6", "\ua7d7"), ("\ua7d8", "\ua7d9"),   # Latin Extended-D
             ("\U00010570", "\U00010597"), ("\U0001057a", "\U000105a1"), ("\U0001057c", "\U000105a3"),
             ("\U0001058a", "\U000105b1"), ("\U0001058c", "\U000105b3"), ("\U00010592", "\U000105b9"),
             ("\U00010594", "\U000105bb"), ("\U00010595", "\U000105bc")]                                 # Vithkuqi, every sub-range's ends
    for up, lo in cases:
        text = "x%sy %s" % (up, up)
        want = ["x%sy" % lo, lo]
        assert Hst.tokenize(text) == want, hex(ord(up))
        assert W.basic_whitespace_lower_tokenizer(text) == want
    # the gaps inside the Vithkuqi block stay as they are
    for cp in (0x1057B, 0x1058B, 0x10593):
        assert Hst.tokenize(chr(cp)) == [chr(cp)]


def test_tokenizer_unicode_folding_and_spaces():
    # row_matcher_test.go:38-41: Kelvin sign folds to 'k' (3 bytes -> 1), U+0130 folds to 'i', NBSP splits
    assert Hst.tokenize("Kelvin") == ["kelvin"]
    assert Hst.tokenize("İstanbul") == ["istanbul"]
    assert Hst.tokenize("a b c　de") == ["a", "b", "c", "d", "e"]
    assert Hst.tokenize("ÀÉÎ ΣΑΣ Ünï") == ["àéî", "σασ", "ünï"]  # simple mapping: no final-sigma rule
    # invalid UTF-8 bytes become U+FFFD, exactly as strings.ToLower re-encodes them
    assert Hst.tokenize(b"ab\xffcd \xc3(") == ["ab�cd", "�("]
    rng = np.random.default_rng(3)
    alphabet = list("aZ \t\n9-é  ΩKİ日本") + ["\U0001F60A"]
    for _ in range(300):
        s = "".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=rng.integers(0, 30)))
        assert Hst.tokenize(s) == W.basic_whitespace_lower_tokenizer(s), repr(s)


# (row, [(kind, args, expected)]) — TestJSONMatching
JSON_MATCHING = [
    ('{"user": {"name": "John", "age": 30}}', [
        ("F", ("user.name",), True), ("F", ("user.age",), True), ("F", ("user",), True),
        ("F", ("user.email",), False), ("F", ("nothere",), False)]),
    ('{"items": [{"name": "Item1", "price": 10}, {"name": "Item2", "price": 20}]}', [
        ("F", ("items.name",), True), ("F", ("items.price",), True), ("F", ("items.category",), False)]),
    ('{"orders": [{"items": [{"name": "A"}, {"name": "B"}]}, {"items": [{"name": "C"}]}]}', [
        ("F", ("orders.items.name",), True)]),
    ('{"user": {"name": "John Doe", "age": 30}}', [
        ("T", ("john",), True), ("T", ("doe",), True), ("T", ("30",), True), ("T", ("jane",), False)]),
    ('{"items": [{"name": "Item1"}, {"name": "Item2"}, {"name": "Item3"}]}', [
        ("T", ("item1",), True), ("T", ("item2",), True), ("T", ("item3",), True), ("T", ("item4",), False)]),
    ('{"user": {"name": "John Doe", "role": "admin"}}', [
        ("FT", ("user.name", "john"), True), ("FT", ("user.name", "doe"), True), ("FT", ("user.role", "admin"), True),
        ("FT", ("user.name", "admin"), False), ("FT", ("user.role", "john"), False), ("FT", ("user.email", "test"), False)]),
    ('{"users": [{"name": "John"}, {"name": "Jane"}], "tags": ["admin", "user"]}', [
        ("FT", ("users.name", "john"), True), ("FT", ("users.name", "jane"), True), ("FT", ("tags", "admin"), True),
        ("FT", ("users.name", "bob"), False), ("FT", ("users.name", "alice"), False)]),
    ('{"groups": [{"users": [{"name": "John"}, {"name": "Jane"}]}, {"users": [{"name": "Bob"}]}]}', [
        ("FT", ("groups.users.name", "john"), True), ("FT", ("groups.users.name", "jane"), True),
        ("FT", ("groups.users.name", "bob"), True), ("FT", ("groups.users.name", "alice"), False)]),
    ('{"items": [{"name": "Item1", "category": "electronics"}, {"name": "Item2", "category": "books"}]}', [
        ("FT", ("items.name", "item1"), True), ("FT", ("items.name", "item2"), True),
        ("FT", ("items.category", "electronics"), True), ("FT", ("items.category", "books"), True),
        ("FT", ("items.name", "item3"), False), ("FT", ("items.category", "furniture"), False)]),
    ('{"tags": [{"type": "admin"}, {"type": "user"}, {"type": "admin"}]}', [
        ("FT", ("tags.type", "admin"), True), ("FT", ("tags.type", "user"), True), ("FT", ("tags.type", "guest"), False)]),
    ('{"records": [{"id": 1, "active": true}, {"id": 2, "active": false}]}', [
        ("FT", ("records.id", "1"), True), ("FT", ("records.id", "2"), True),
        ("FT", ("records.active", "true"), True), ("FT", ("records.active", "false"), True)]),
    ('{"user": {"name": "John", "tags": [{"type": "admin"}, {"role": "user"}]}}', [
        ("FT", ("user.name", "john"), True), ("FT", ("user.tags.type", "admin"), True), ("FT", ("user.tags.role", "user"), True),
        ("FT", ("user.tags.type", "user"), False), ("FT", ("user.tags.role", "admin"), False)]),
    # no_false_negatives_test.go regressions (rows as Go's json.Marshal emits them)
    ('{"id":1,"user_id":1234567}', [("FT", ("user_id", "1234567"), True), ("T", ("1234567",), True), ("T", ("1.234567e+06",), False)]),
    ('{"big":9007199254740993,"id":2}', [("FT", ("big", "9007199254740993"), True), ("FT", ("big", "9007199254740992"), False)]),
    ('{"a.b":"hello","id":1}', [("F", ("a.b",), True), ("F", ("a",), True), ("FT", ("a.b", "hello"), True), ("FT", ("a", "hello"), False)]),
    ('{"a":{"b":"world"},"id":2}', [("F", ("a.b",), True), ("F", ("a",), True), ("FT", ("a.b", "world"), True)]),
    ('{".a":"xyz","id":4}', [("F", ("",), False), ("F", (".a",), True), ("FT", (".a", "xyz"), True)]),
    ('{"a*":1,"id":2}', [("F", ("a*",), True), ("FT", ("a*", "1"), True), ("F", ("ab",), False)]),
    ('{"ab":"x","id":1}', [("F", ("a*",), False), ("F", ("ab",), True)]),
    ('{"back\\\\slash":"v","id":3,"q?x":"y"}', [("F", ("back\\slash",), True), ("FT", ("q?x", "y"), True)]),
    ('{"id":4,"n":null}', [("F", ("n",), True), ("FT", ("n", "null"), False), ("T", ("null",), False)]),
    ('{"id":2,"ts":"2020-01-02T03:04:05Z"}', [("FT", ("ts", "2020-01-02t03:04:05z"), True), ("FT", ("ts", "2020-01-02T03:04:05Z"), False)]),
    ('{"data":"aGk=","id":3}', [("FT", ("data", "agk="), True)]),
    ('{"id":1,"latency":1500000}', [("T", ("1500000",), True), ("T", ("1.5e+06",), False)]),
    ('{"m":"\\u003chtml\\u003e\\u0026amp; x"}', [("T", ("<html>&amp;",), True), ("FT", ("m", "x"), True)]),   # json.Marshal HTML escaping
    ('{"name":"ALICE Smith"}', [("T", ("alice",), True), ("T", ("ALICE",), False)]),                                  # targets never lowercased
]


def _expr(kind, args):
    return {"F": Q.Field, "T": Q.Token, "FT": Q.FieldToken}[kind](*args)


@pytest.mark.parametrize("row,cases", JSON_MATCHING)
def test_json_matching_tables(row, cases):
    for kind, args, expected in cases:
        e = _expr(kind, args)
        assert W.matches_bloom_expression(row, e) == expected, (row, kind, args)     # oracle pinned on the table
        assert Hst.match_row(e, row.encode()) == expected, (row, kind, args)        # product host mirror


def test_query_test_bloom_semantics_rows():
    # query_test.go:91-111
    e = Q.And(Q.Or(Q.Field("user.name"), Q.FieldToken("service", "payment")), Q.Token("timeout"))
    rows = [('{"user":{"name":"dan"},"message":"connection timeout"}', True),
            ('{"service":"payment","message":"a timeout happened"}', True),
            ('{"service":"payment","message":"all good"}', False),
            ('{"other":"x","message":"timeout"}', False)]
    for row, want in rows:
        assert Hst.match_row(e, row.encode()) == want
        assert W.matches_bloom_expression(row, e) == want
    assert Hst.match_row(None, b'{"a":1}') is True
    assert Hst.match_row(Q.And(), b'{"a":1}') is True and Hst.match_row(Q.Or(), b'{"a":1}') is False
    assert Hst.match_row({"ExpressionType": "CONDITION", "Condition": None}, b'{"a":1}') is True
    assert Hst.match_row({"ExpressionType": "WAT"}, b'{"a":1}') is False


def test_field_token_pairs_not_joined_key():
    # row_matcher.go:296-301 / TestFieldTokenJoinedKeyCollisionPinned: the row matcher compares (path, token)
    # pairs; the bloom key is the plain join.  {"a":"b::c"} has pair ("a","b::c"); FieldToken("a::b","c") shares
    # the joined key "a::b::c" but is NOT a row-level match.
    row = b'{"a":"b::c"}'
    assert Hst.match_row(Q.FieldToken("a", "b::c"), row) is True
    assert Hst.match_row(Q.FieldToken("a::b", "c"), row) is False
    s = Hst.EntrySets()
    s.index_row(row)
    assert "a::b::c" in s.as_python_sets()[2]


WORDS = ["alpha", "BETA", "Gamma delta", "a.b", "q?x", "héllo", "日本語", "<html>&amp;", "x::y", "tab\there", "",
         "  spaced  out ", "KK"]
KEYS = ["id", "user", "name", "a.b", "a*", "q?x", "back\\slash", "héllo", "日本語", "tags", ".lead", "trail.", "x..y", "",
        "a::b"]


def _random_value(rng, depth):
    r = rng.random()
    if depth < 3 and r < 0.2:
        return {KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, depth + 1) for _ in range(rng.integers(0, 4))}
    if depth < 3 and r < 0.35:
        return [_random_value(rng, depth + 1) for _ in range(rng.integers(0, 4))]
    if r < 0.5:
        return [0, 1, -5, 42, 1234567, 2 ** 53 + 1, 2 ** 64 - 1, -2 ** 63][rng.integers(0, 8)]
    if r < 0.6:
        return [0.5, -1.25, 3.0e10, 1e-7][rng.integers(0, 4)]
    if r < 0.68:
        return [None, True, False][rng.integers(0, 3)]
    return WORDS[rng.integers(0, len(WORDS))]


def go_marshal(obj) -> bytes:
    """Go's json.Marshal of map[string]any: sorted keys, no spaces, HTML-escaped <, >, &, U+2028/9."""
    s = json.dumps(obj, sort_keys=True, separators=(",", ":"), ensure_ascii=False)
    for a, b in (("<", "\\u003c"), (">", "\\u003e"), ("&", "\\u0026"), (" ", "\\u2028"), (" ", "\\u2029")):
        s = s.replace(a, b)
    return s.encode()


def test_regex_field_guard_keeps_the_boolean_shape():
    """RegexFieldGuardBloomQuery (query.go:651-707) and pruneBloomQuery = AndBloomQueries(bloom, guard) (query_exec.go:220),
    host mirror against the Python restatement; shape test after query_builder_test.go:214-260."""
    regex = Q.RegexOr(Q.FieldRegex("service", "^pay"), Q.RegexAnd(Q.FieldRegex("level", "^error$"), Q.FieldRegex("message", "timeout")))
    guard = Hst.prune_query(None, regex)
    assert guard == Q.regex_field_guard_bloom_query(regex)
    assert guard["ExpressionType"] == "OR" and len(guard["Children"]) == 2
    assert guard["Children"][0]["Condition"] == {"Type": "FIELD", "Field": "service", "Token": ""}
    assert [c["Condition"]["Field"] for c in guard["Children"][1]["Children"]] == ["level", "message"]
    # AndBloomQueries: nil sides pass through, two real sides are And-ed WITH same-type flattening (query.go:709-718, :600-610)
    bloom = Q.And(Q.Field("user.name"), Q.Token("timeout"))
    both = Hst.prune_query(bloom, Q.RegexAnd(Q.FieldRegex("a", "x"), Q.FieldRegex("b", "y")))
    want = Q.prune_bloom_query(bloom, Q.RegexAnd(Q.FieldRegex("a", "x"), Q.FieldRegex("b", "y")))
    strip = lambda e: ({"ExpressionType": e["ExpressionType"], "Children": [strip(c) for c in e.get("Children", [])]} if e["ExpressionType"] != "CONDITION"
                       else {"ExpressionType": "CONDITION", "Condition": {k: v for k, v in (e["Condition"] or {}).items() if v != ""}})
    assert strip(both) == strip(want)
    assert [c["Condition"]["Type"] for c in both["Children"]] == ["FIELD", "TOKEN", "FIELD", "FIELD"]      # flattened into one AND
    assert Hst.prune_query(bloom, None) is not None and Hst.prune_query(None, None) is None
    # a nil regex condition contributes nothing; an unknown node type makes the whole guard nil
    assert Hst.prune_query(None, {"ExpressionType": "AND", "Children": [{"ExpressionType": "CONDITION", "Condition": None}]}) == \
        {"ExpressionType": "AND", "Children": []}
    assert Hst.prune_query(None, {"ExpressionType": "XOR", "Children": []}) is None


def test_regex_row_matching_tables():
    """FieldRegex semantics (row_matcher.go:440-476, :519-573): the pattern sees the text of every primitive at or beneath
    the field path.  Rows and verdicts of query_test.go:119-128 and tokenizer_test.go:192-215."""
    regex = Q.RegexOr(Q.RegexAnd(Q.FieldRegex("message", "timeout|retry"), Q.FieldRegex("level", "^err")), Q.FieldRegex("service", "^pay"))
    for row, want in [(b'{"message":"retry now","level":"error"}', True), (b'{"service":"payments"}', True),
                      (b'{"message":"retry now","level":"info"}', False)]:
        assert Hst.match_row_regex(regex, row) == want, row
    nested = Q.RegexAnd(Q.FieldRegex("users.name", "(?i)^jo"), Q.RegexOr(Q.FieldRegex("users.active", "^true$"), Q.FieldRegex("users.id", "^2$")))
    assert Hst.match_row_regex(nested, b'{"users":[{"id":1,"name":"John","active":true},{"id":2,"name":"Jane","active":false}]}')
    assert not Hst.match_row_regex(nested, b'{"users":[{"id":3,"name":"Alice","active":false}]}')
    # beneath the path, raw number literals, null never, empty field constant-false, nil condition true, empty Or false
    assert Hst.match_row_regex(Q.FieldRegex("a", "^1500000$"), b'{"a":{"b":[1500000]}}')
    assert not Hst.match_row_regex(Q.FieldRegex("a", "."), b'{"a":null,"ab":"x"}')
    assert not Hst.match_row_regex(Q.FieldRegex("", "."), b'{"a":"x"}')
    assert Hst.match_row_regex({"ExpressionType": "CONDITION", "Condition": None}, b'{"a":"x"}')
    assert not Hst.match_row_regex({"ExpressionType": "OR", "Children": []}, b'{"a":"x"}')
    with pytest.raises(Hst.HostError):
        Hst.match_row_regex(Q.FieldRegex("message", "[unterminated("), b'{"message":"x"}')


def test_index_row_matches_walker_oracle_on_random_rows():
    # the property generator of no_false_negatives_test.go:398-459, re-seeded (any seeded RNG does)
    rng = np.random.default_rng(7)
    union = (set(), set(), set())
    acc = Hst.EntrySets()
    for _ in range(150):
        row = go_marshal({KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))})
        s = Hst.EntrySets()
        s.index_row(row)
        want = W.index_row(row)
        got = s.as_python_sets()
        assert got == want, row
        assert s.counts() == tuple(len(x) for x in want)
        s.union_into(acc)
        for u, w in zip(union, want):
            u |= w
    assert acc.as_python_sets() == union          # unionInto (ingest.go:105-115)
    assert acc.counts() == tuple(len(u) for u in union)


def test_index_row_rejects_malformed_json():
    s = Hst.EntrySets()
    with pytest.raises(Hst.HostError):
        s.index_row(b'{"a": [1, 2')
    s = Hst.EntrySets()
    s.index_row(b'{"dup": 1, "dup": {"x": "y"}}')      # duplicate keys: both visited, in order
    assert s.as_python_sets()[0] == {"dup", "dup.x"}


def test_batch_lowering_matches_python_compile():
    rng = np.random.default_rng(9)
    from tests.helpers import random_expression
    vocab = ["tok%d" % i for i in range(50)]
    exprs = [None] + [random_expression(rng, vocab, None) for _ in range(200)]
    cb = Q.compile_queries(exprs)
    ops, poff, kinds = cb.arrays()
    strings, hkinds, hops, hpoff = Hst.HostBatch(exprs).export()
    assert strings == cb.term_strings
    assert np.array_equal(hkinds, kinds) and np.array_equal(hops, ops) and np.array_equal(hpoff, poff)


def test_section_codec_matches_oracle_bytes():
    from oracle import oracle as O
    f = O.build_sized(["user.name", "user.age"], 0.01)
    t = O.build_sized(["tok%d" % i for i in range(1000)], 0.001)
    for filters in ([f, t, None], [None, None, None], [f, None, t], [None, t, None]):
        want = O.encode_filter_section(filters)
        got = Hst.section_encode([None if x is None else (x.m, x.k, x.words) for x in filters])
        assert got == want
        back = Hst.section_parse(got)
        for a, b in zip(filters, back):
            assert (a is None) == (b is None)
            if a is not None:
                assert (a.m, a.k) == b[:2] and np.array_equal(a.words, b[2])
    sec = bytearray(O.encode_filter_section([f, t, None]))
    sec[7] ^= 1
    with pytest.raises(Hst.HostError) as e:
        Hst.section_parse(bytes(sec))
    assert e.value.code == -2          # ErrInvalidHash
    assert Hst.crc32c(b"123456789") == 0xE3069283
    # crafted headers under a CORRECT checksum: m so large that (m + 63) / 64 wraps (found by tools/fuzz_sections.py: such a
    # section used to parse as "m = 2^64 - 1 with zero words"), a bitset length that wraps, k = 0, k beyond the cap
    import struct

    def crafted(m, k, blen, n_words):
        body = bytes([1]) + struct.pack("<I", 24 + 8 * n_words) + struct.pack(">QQQ", m, k, blen) + b"\xAA" * (8 * n_words)
        return body + struct.pack("<I", Hst.crc32c(body))
    assert Hst.section_parse(crafted(64, 3, 64, 1))[0][:2] == (64, 3)
    for m, k, blen in ((2 ** 64 - 1, 3, 64), (2 ** 64 - 63, 3, 64), (2 ** 64 - 64, 3, 64), (64, 3, 2 ** 64 - 1), (64, 0, 64), (64, 1025, 64),
                       (65, 3, 64), (0, 3, 64)):
        with pytest.raises(Hst.HostError) as e:
            Hst.section_parse(crafted(m, k, blen, 1))
        assert e.value.code == -5, (m, k, blen)


def test_host_symbols_exported():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "bloomsearch_host.h")).read()
    declared = set(re.findall(r"BSG_API\s+[\w\s\*]+?\b(bs[he]_\w+)\s*\(", hdr))
    assert declared == set(Hst.HOST_EXPORTS), declared ^ set(Hst.HOST_EXPORTS)
    L = Hst.lib()
    for n in declared:
        assert hasattr(L, n)
