"""The tree-walking oracle evaluator (oracle.evaluate_tree*: evaluateBloomFilters / evaluateBloomExpression /
evaluateBloomCondition restated over the expression TREE, query_exec.go:75-159) pinned on the reference's own 8-verdict
fixture, and used as the checker of the product's LOWERING (bloomsearch_amd.query.compile_queries -> postfix programs):
the postfix form is what bsg_probe consumes, so a wrong lowering of a nil / unknown / empty node shows up here, on the
CPU, without a GPU in the loop."""
import json
import os

import numpy as np

from bloomsearch_amd import query as Q
from oracle import oracle as O
from tests import helpers as H

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bloom_vectors.json")))


def test_tree_evaluator_reproduces_the_reference_fixture():
    """TestEvaluateBloomFilters (bloom_tree_engine_test.go:357-442): three (959, 7) filters of two entries, 8 verdicts."""
    fx = G["evaluate_bloom_filters_fixture"]
    filters = O.parse_filter_section(bytes.fromhex(fx["section_hex"]))
    for case in fx["cases"]:
        assert O.evaluate_tree(filters, case["expression"]) == case["expected"], case["name"]
    # nil filters cannot disqualify (query_exec.go:137-151)
    assert O.evaluate_tree([None, None, None], Q.And(Q.Field("zz"), Q.Token("zz"), Q.FieldToken("zz", "zz")))
    # unknown expression / condition type => false; nil condition => true; And() => true; Or() => false
    assert not O.evaluate_tree(filters, {"ExpressionType": "XOR", "Children": []})
    assert not O.evaluate_tree(filters, {"ExpressionType": "CONDITION", "Condition": {"Type": "BOGUS"}})
    assert O.evaluate_tree(filters, {"ExpressionType": "CONDITION", "Condition": None})
    assert O.evaluate_tree(filters, Q.And()) and not O.evaluate_tree(filters, Q.Or())


def test_blockwise_tree_walk_equals_scalar_tree_walk_and_postfix_oracle():
    rng = np.random.default_rng(77)
    plan, blocks_str, vocab = H.make_random_arena(rng, 70, max_tokens=400)
    words = H.oracle_words(plan)
    desc = plan.desc.view(O.DESC_DTYPE)
    exprs = [None] + [H.random_expression(rng, vocab, None) for _ in range(400)]
    for b in rng.integers(0, 70, size=30):
        f, t, ft = blocks_str[b]
        if t:
            fld, tok = ft[rng.integers(0, len(ft))].split("::", 1)
            exprs.append(Q.And(Q.Field(f[0]), Q.Or(Q.Token(t[rng.integers(0, len(t))]), Q.FieldToken(fld, tok))))
    tree = O.survivors_tree(words, desc, exprs)
    # scalar recursion with short circuits, block by block, on a sample of queries
    for q in rng.choice(len(exprs), size=60, replace=False):
        for b in range(70):
            want = O.evaluate_tree(O.block_filters(words, desc, b), exprs[q])
            assert bool((int(tree[q, b >> 6]) >> (b & 63)) & 1) == want, (q, b)
    # the product's lowering, evaluated by the oracle's postfix interpreter, must give the same survivor sets
    cb = Q.compile_queries(exprs)
    ops, poff, _ = cb.arrays()
    postfix = O.probe_batch(words, desc, H.oracle_terms(cb).view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(tree, postfix)
    assert tree.any() and not tree.all()


def test_cpp_host_lowering_is_checked_by_the_tree_oracle():
    """csrc/host/expression.hpp lowers the same trees for the engine mirror; its postfix programs, run by the oracle's
    interpreter, must reproduce the tree-walking oracle's survivor sets (test_host_tables.py only compares the two
    lowerings with each other)."""
    from bloomsearch_amd import host as Hst
    rng = np.random.default_rng(78)
    plan, _, vocab = H.make_random_arena(rng, 40, max_tokens=300)
    words = H.oracle_words(plan)
    desc = plan.desc.view(O.DESC_DTYPE)
    exprs = [None] + [H.random_expression(rng, vocab, None) for _ in range(300)]
    strings, kinds, ops, poff = Hst.HostBatch(exprs).export()
    terms = np.zeros(len(strings), dtype=O.TERM_DTYPE)
    for i, s in enumerate(strings):
        terms["h"][i] = O.base_hashes(s)
        terms["kind"][i] = kinds[i]
    assert np.array_equal(O.probe_batch(words, desc, terms, ops, poff), O.survivors_tree(words, desc, exprs))
