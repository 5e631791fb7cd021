"""CPU ORACLE (test infrastructure) for the row side of the path: a pure-Python
restatement of the reference's row enumeration, tokenizer, entry sets and
set-based row matcher.  Small cases only; never imported by bloomsearch_amd/.

Follows (reference file:line):
  forEachPathValue / walkPathValues / emitKeyPrefixPaths .... tokenizer.go:51-113
  leafTokenInput ............................................. tokenizer.go:120-133
  BasicWhitespaceLowerTokenizer = Fields(ToLower(v)) ......... tokenizer.go:141-143
  bloomEntrySets.indexRow / addFieldToken .................... ingest.go:55-102
  buildRowMatchSets / matchesBloomExpression ................. tokenizer.go:236-330
Pinned by the language-neutral tables of tokenizer_test.go:10-190,
query_test.go:91-111 and no_false_negatives_test.go:103-321 (tests/test_host_tables.py, tests/test_ingest_gpu.py, tests/test_match_gpu.py).

gjson (v1.18.0, un-vendored) semantics relied on: Parse + ForEach visit object
members in document order including duplicate keys; key.String()/value.Str are
the JSON-unescaped text; value.Raw of a number is its literal text.
"""
from __future__ import annotations

import json

DELIM = "."

# Go unicode.IsSpace (White_Space property): the set strings.Fields splits on
_SPACE = {chr(c) for c in (0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000)} | \
    {chr(c) for c in range(0x2000, 0x200B)}


class _Num:
    """A JSON number kept as its raw literal (gjson .Raw; never a float round trip)."""
    __slots__ = ("raw",)

    def __init__(self, raw):
        self.raw = raw


def parse(row: bytes | str):
    if isinstance(row, bytes):
        row = row.decode("utf-8", "surrogatepass")
    return json.loads(row, object_pairs_hook=lambda pairs: ("obj", pairs), parse_int=_Num, parse_float=_Num,
                      parse_constant=_Num)


def _is_obj(v):
    return isinstance(v, tuple) and len(v) == 2 and v[0] == "obj"


def for_each_path_value(value, emit, delimiter: str = DELIM):
    """emit(path, value, is_leaf) — tokenizer.go:51-82."""
    _walk(value, "", delimiter, emit)


def _walk(value, path, delimiter, emit):
    if _is_obj(value):
        if path != "":
            emit(path, value, False)
        for key, child in value[1]:
            child_path = key if path == "" else path + delimiter + key
            _emit_key_prefix_paths(path, key, delimiter, emit)
            _walk(child, child_path, delimiter, emit)
    elif isinstance(value, list):
        if path != "":
            emit(path, value, False)
        for child in value:
            _walk(child, path, delimiter, emit)      # array elements share the array's path
    elif path != "":
        emit(path, value, True)


def _emit_key_prefix_paths(parent, key, delimiter, emit):
    """tokenizer.go:93-113: every delimiter-split prefix of the key is a field-existence path."""
    if delimiter == "":
        return
    split_at = 0
    while True:
        idx = key.find(delimiter, split_at)
        if idx == -1:
            return
        split_at = idx
        prefix = key[:split_at]
        if parent != "":
            prefix = parent + delimiter + prefix
        if prefix != "":
            emit(prefix, None, False)
        split_at += len(delimiter)


def leaf_token_input(value):
    """tokenizer.go:120-133 -> text or None."""
    if isinstance(value, str):
        return value
    if isinstance(value, _Num):
        return value.raw
    if value is True:
        return "true"
    if value is False:
        return "false"
    return None


# Simple lowercase mappings Unicode 14.0 added (UnicodeData.txt 14.0.0, field 13), which this interpreter's Unicode 13.0
# data lacks and every Go >= 1.21 (Unicode 15.0) has.  Transcribed independently of tools/gen_unicode_tables.py:
#   2C2F -> 2C5F (Glagolitic caudate chrivi); A7C0 -> A7C1, A7D0 -> A7D1, A7D6 -> A7D7, A7D8 -> A7D9 (Latin Extended-D);
#   Vithkuqi capitals 10570-1057A, 1057C-1058A, 1058C-10592, 10594-10595 -> +0x27.
_UNICODE_14_LOWER = {0x2C2F: 0x2C5F, 0xA7C0: 0xA7C1, 0xA7D0: 0xA7D1, 0xA7D6: 0xA7D7, 0xA7D8: 0xA7D9}
for _a, _b in ((0x10570, 0x1057A), (0x1057C, 0x1058A), (0x1058C, 0x10592), (0x10594, 0x10595)):
    for _cp in range(_a, _b + 1):
        _UNICODE_14_LOWER[_cp] = _cp + 0x27


def _to_lower_rune(ch: str) -> str:
    """Go unicode.ToLower: SIMPLE case mapping (one rune -> one rune).  Python's str.lower() is the
    full mapping; for a single character they differ only for U+0130 (full: 'i' + U+0307)."""
    if ch == "\u0130":
        return "i"
    d = _UNICODE_14_LOWER.get(ord(ch))
    if d is not None:
        return chr(d)
    lo = ch.lower()
    return lo if len(lo) == 1 else ch


def basic_whitespace_lower_tokenizer(text: str) -> list[str]:
    """strings.Fields(strings.ToLower(text)) for valid UTF-8 text."""
    lowered = "".join(_to_lower_rune(c) for c in text)
    out, cur = [], []
    for c in lowered:
        if c in _SPACE:
            if cur:
                out.append("".join(cur))
                cur = []
        else:
            cur.append(c)
    if cur:
        out.append("".join(cur))
    return out


def index_row(row, sets=None):
    """bloomEntrySets.indexRow (ingest.go:55-89) -> (fields, tokens, field_tokens) sets of str."""
    fields, tokens, field_tokens = sets if sets is not None else (set(), set(), set())

    def emit(path, value, is_leaf):
        fields.add(path)
        if not is_leaf:
            return
        text = leaf_token_input(value)
        if text is None:
            return
        for tok in basic_whitespace_lower_tokenizer(text):
            tokens.add(tok)
            field_tokens.add(path + "::" + tok)

    for_each_path_value(parse(row), emit)
    return fields, tokens, field_tokens


def matches_bloom_expression(row, expression) -> bool:
    """testJSONForBloomQuery: buildRowMatchSets + matchesBloomExpression (tokenizer.go:236-330)."""
    fields, tokens, field_tokens = index_row(row)

    def cond(c):
        t = c.get("Type")
        if t == "FIELD":
            return c.get("Field", "") in fields
        if t == "TOKEN":
            return c.get("Token", "") in tokens
        if t == "FIELD_TOKEN":
            return (c.get("Field", "") + "::" + c.get("Token", "")) in field_tokens
        return False

    def ev(e):
        if e is None:
            return True
        et = e.get("ExpressionType")
        if et == "CONDITION":
            return True if e.get("Condition") is None else cond(e["Condition"])
        kids = e.get("Children") or []
        if et == "OR":
            return any(ev(k) for k in kids)
        if et == "AND":
            return all(ev(k) for k in kids)
        return False

    return ev(expression)
