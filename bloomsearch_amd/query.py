"""Bloom query algebra: a host-side mirror of the reference's BloomExpression tree.

Mirrors query.go:478-610 (BloomCondition / BloomExpression, Field / Token /
FieldToken / And / Or with same-type flattening, flattenExpressions :600-610)
and AndBloomQueries (:709-718).  Expressions are plain dicts in the exact JSON
shape of the reference's exported structs, so a tree serialised by a Go
MetaStore/tool loads unchanged.

compile_queries() lowers a batch of trees to the C-ABI form (bloomgpu.h):
distinct terms (the probed strings + filter kind) and one postfix program per
query, following evaluateBloomExpression / evaluateBloomCondition
(query_exec.go:89-159) case by case.
"""
from __future__ import annotations

import numpy as np

from ._lib import KIND_FIELD, KIND_FIELD_TOKEN, KIND_TOKEN, OP_AND, OP_FALSE, OP_OR, OP_TERM, OP_TRUE, op

BLOOM_FIELD, BLOOM_TOKEN, BLOOM_FIELD_TOKEN = "FIELD", "TOKEN", "FIELD_TOKEN"
EXPR_CONDITION, EXPR_AND, EXPR_OR = "CONDITION", "AND", "OR"


def Field(field: str) -> dict:
    return {"ExpressionType": EXPR_CONDITION, "Condition": {"Type": BLOOM_FIELD, "Field": field, "Token": ""}}


def Token(token: str) -> dict:
    return {"ExpressionType": EXPR_CONDITION, "Condition": {"Type": BLOOM_TOKEN, "Field": "", "Token": token}}


def FieldToken(field: str, token: str) -> dict:
    return {"ExpressionType": EXPR_CONDITION,
            "Condition": {"Type": BLOOM_FIELD_TOKEN, "Field": field, "Token": token}}


def _flatten(expressions, expression_type):
    # query.go:600-610: a child of the same type with no Condition is spliced in
    out = []
    for e in expressions:
        if e.get("ExpressionType") == expression_type and e.get("Condition") is None:
            out.extend(e.get("Children") or [])
        else:
            out.append(e)
    return out


def And(*expressions) -> dict:
    return {"ExpressionType": EXPR_AND, "Children": _flatten(expressions, EXPR_AND)}


def Or(*expressions) -> dict:
    return {"ExpressionType": EXPR_OR, "Children": _flatten(expressions, EXPR_OR)}


def and_bloom_queries(left, right):
    """AndBloomQueries (query.go:709-718) on bare expressions (None == nil query)."""
    if left is None:
        return right
    if right is None:
        return left
    return And(left, right)


# ---- regex query tree (query.go:520-538, :612-649) and its bloom field guard (:651-707) ----
def FieldRegex(field: str, pattern: str) -> dict:
    return {"ExpressionType": "CONDITION", "Condition": {"Field": field, "Pattern": pattern}}


def _flatten_regex(expressions, expression_type):
    out = []
    for e in expressions:
        if e.get("ExpressionType") == expression_type and e.get("Condition") is None:
            out.extend(e.get("Children") or [])
        else:
            out.append(e)
    return out


def RegexAnd(*expressions) -> dict:
    return {"ExpressionType": "AND", "Children": _flatten_regex(expressions, "AND")}


def RegexOr(*expressions) -> dict:
    return {"ExpressionType": "OR", "Children": _flatten_regex(expressions, "OR")}


def regex_field_guard_bloom_query(regex_expression):
    """RegexFieldGuardBloomQuery (query.go:698-707) on bare expressions: every regex condition becomes Field(cond.Field);
    And / Or keep their node around whatever children translate (no flattening); anything else is nil."""
    if regex_expression is None:
        return None
    t = regex_expression.get("ExpressionType")
    if t == "CONDITION":
        c = regex_expression.get("Condition")
        return None if c is None else Field(c.get("Field", ""))
    if t in ("AND", "OR"):
        kids = [g for g in (regex_field_guard_bloom_query(c) for c in regex_expression.get("Children") or []) if g is not None]
        return {"ExpressionType": EXPR_AND if t == "AND" else EXPR_OR, "Children": kids}
    return None


def prune_bloom_query(bloom_expression, regex_expression):
    """pruneBloomQuery (query_exec.go:220)."""
    return and_bloom_queries(bloom_expression, regex_field_guard_bloom_query(regex_expression))


def make_field_token_key(field: str, token: str) -> str:
    """makeFieldTokenKey (tokenizer.go:509-511): plain concatenation, no escaping."""
    return field + "::" + token


def term_of(condition: dict):
    """(kind, probed string) for a known condition type, else None (query_exec.go:134-157)."""
    t = condition.get("Type")
    if t == BLOOM_FIELD:
        return KIND_FIELD, condition.get("Field", "")
    if t == BLOOM_TOKEN:
        return KIND_TOKEN, condition.get("Token", "")
    if t == BLOOM_FIELD_TOKEN:
        return KIND_FIELD_TOKEN, make_field_token_key(condition.get("Field", ""), condition.get("Token", ""))
    return None


class CompiledBatch:
    """Distinct terms + per-query postfix programs for a batch of queries."""

    def __init__(self):
        self.term_strings: list[bytes] = []
        self.term_kinds: list[int] = []
        self._index: dict = {}
        self.prog_ops: list[int] = []
        self.prog_off: list[int] = [0]

    def _term(self, kind: int, s: str) -> int:
        key = (kind, s)
        i = self._index.get(key)
        if i is None:
            i = len(self.term_strings)
            self._index[key] = i
            self.term_strings.append(s.encode("utf-8", "surrogatepass"))
            self.term_kinds.append(kind)
        return i

    def _emit(self, e) -> None:
        if e is None:                       # nil expression => true (query_exec.go:96-98)
            self.prog_ops.append(op(OP_TRUE))
            return
        et = e.get("ExpressionType")
        if et == EXPR_CONDITION:
            cond = e.get("Condition")
            if cond is None:                # nil condition => true (:101-104)
                self.prog_ops.append(op(OP_TRUE))
                return
            t = term_of(cond)
            if t is None:                   # unknown condition type => false (:155-156)
                self.prog_ops.append(op(OP_FALSE))
            else:
                self.prog_ops.append(op(OP_TERM, self._term(*t)))
        elif et in (EXPR_AND, EXPR_OR):
            kids = e.get("Children") or []
            for c in kids:
                self._emit(c)
            self.prog_ops.append(op(OP_AND if et == EXPR_AND else OP_OR, len(kids)))
        else:                               # unknown expression type => false (:122-123)
            self.prog_ops.append(op(OP_FALSE))

    def add_query(self, expression) -> None:
        """expression None == nil BloomQuery / nil Expression => no ops => true (:81-83)."""
        if expression is not None:
            self._emit(expression)
        self.prog_off.append(len(self.prog_ops))

    @property
    def n_queries(self) -> int:
        return len(self.prog_off) - 1

    def arrays(self):
        return (np.asarray(self.prog_ops, dtype=np.uint32), np.asarray(self.prog_off, dtype=np.uint32),
                np.asarray(self.term_kinds, dtype=np.uint32))


def compile_queries(expressions) -> CompiledBatch:
    cb = CompiledBatch()
    for e in expressions:
        cb.add_query(e)
    return cb


class CompiledMatcher:
    """One expression for the device row matcher (include/bloomgpu.h bsg_match_rows): its conditions — field and token
    strings kept APART, because FieldToken is a (path, token) pair at one leaf, not the joined bloom key
    (row_matcher.go:587) — and the postfix program over condition indices."""

    def __init__(self, expression):
        self.kinds: list[int] = []
        self.fields: list[bytes] = []
        self.tokens: list[bytes] = []
        self.prog_ops: list[int] = []
        if expression is not None:
            self._emit(expression)

    def _emit(self, e) -> None:
        if e is None:
            self.prog_ops.append(op(OP_TRUE))
            return
        et = e.get("ExpressionType")
        if et == EXPR_CONDITION:
            cond = e.get("Condition")
            if cond is None:                # nil condition => true (row_matcher.go:257-290)
                self.prog_ops.append(op(OP_TRUE))
                return
            t = cond.get("Type")
            if t not in (BLOOM_FIELD, BLOOM_TOKEN, BLOOM_FIELD_TOKEN):
                self.prog_ops.append(op(OP_FALSE))
                return
            self.prog_ops.append(op(OP_TERM, len(self.kinds)))
            self.kinds.append({BLOOM_FIELD: KIND_FIELD, BLOOM_TOKEN: KIND_TOKEN, BLOOM_FIELD_TOKEN: KIND_FIELD_TOKEN}[t])
            self.fields.append(cond.get("Field", "").encode("utf-8", "surrogatepass"))
            self.tokens.append(cond.get("Token", "").encode("utf-8", "surrogatepass"))
        elif et in (EXPR_AND, EXPR_OR):
            kids = e.get("Children") or []
            for c in kids:
                self._emit(c)
            self.prog_ops.append(op(OP_AND if et == EXPR_AND else OP_OR, len(kids)))
        else:
            self.prog_ops.append(op(OP_FALSE))
