"""Committed golden vectors (tests/golden/bloom_vectors.json, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them (guards the oracle against drift).
GPU: the HIP path reproduces them through the C-ABI (hashes, bitsets, wire bytes, survivors)."""
import json
import os

import numpy as np
import pytest

from bloomsearch_amd import query as Q
from bloomsearch_amd._lib import DESC_DTYPE, TERM_DTYPE
from oracle import oracle as O

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bloom_vectors.json")))


def test_oracle_reproduces_golden_vectors():
    for v in G["murmur3_x64_128_public"]["vectors"]:
        assert ["%016x" % x for x in O.murmur3_x64_128(v["data"].encode())] == v["h"]
    for v in G["estimate_parameters"]["vectors"]:
        assert O.estimate_parameters(v["n"], v["p"]) == (v["m"], v["k"])
    for v in G["base_hashes_and_locations"]["vectors"]:
        s = bytes.fromhex(v["hex"])
        h = O.base_hashes(s)
        assert ["%016x" % x for x in h] == v["h"]
        assert [O.location(h, i) % 959 for i in range(11)] == v["loc_m959"]
        assert [O.location(h, i) % 287552 for i in range(11)] == v["loc_m287552"]
        assert [O.location(h, i) % ((1 << 40) + 7) for i in range(11)] == v["loc_m2p40plus7"]
    fx = G["evaluate_bloom_filters_fixture"]
    filters = O.parse_filter_section(bytes.fromhex(fx["section_hex"]))
    assert [f.serialize().hex() for f in filters] == fx["filters_hex"]
    assert all((f.m, f.k) == (959, 7) for f in filters)
    for v in G["crc32c"]["vectors"]:
        assert O.crc32c(v["data"].encode()) == v["crc"]


def test_host_mirror_estimate_parameters_golden():
    from bloomsearch_amd.gpu import estimate_parameters
    for v in G["estimate_parameters"]["vectors"]:
        assert estimate_parameters(v["n"], v["p"]) == (v["m"], v["k"])


@pytest.mark.gpu
def test_gpu_hashes_match_golden(ctx):
    vec = G["base_hashes_and_locations"]["vectors"]
    got = ctx.hash_strings([bytes.fromhex(v["hex"]) for v in vec])
    for row, v in zip(got, vec):
        assert ["%016x" % int(x) for x in row] == v["h"]


@pytest.mark.gpu
def test_gpu_evaluate_bloom_filters_fixture(ctx):
    """TestEvaluateBloomFilters (bloom_tree_engine_test.go:357-442): same three (959,7) filters, same 8 verdicts;
    the filters are rebuilt on the GPU and must serialise to the golden wire bytes."""
    from bloomsearch_amd import host as Hst
    from bloomsearch_amd.arena import entry_sets_from_strings
    fx = G["evaluate_bloom_filters_fixture"]
    sets = entry_sets_from_strings(["user.name", "user.age"], ["alice", "30"], ["user.name::alice", "user.age::30"])
    desc = np.zeros(3, dtype=DESC_DTYPE)
    fstart, blobs, lens = [0], [], []
    for c in range(3):
        desc[c] = (c * 16, 959, 7, 0)          # NewWithEstimates(100, 0.01), not sized for the 2 entries
        blobs.append(sets[c][0]); lens.append(sets[c][1])
        fstart.append(fstart[-1] + len(sets[c][1]))
    ln = np.concatenate(lens)
    off = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint32)
    words = ctx.build(np.concatenate(blobs), off, np.asarray(fstart, dtype=np.uint32), desc, 48)
    sec = Hst.section_encode([(959, 7, words[c * 16: c * 16 + 15].copy()) for c in range(3)])
    assert sec.hex() == fx["section_hex"]
    exprs = [c["expression"] for c in fx["cases"]]
    cb = Q.compile_queries(exprs)
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    aid = ctx.arena_load(words, desc)
    got = ctx.probe(aid, 1, terms, ops, poff)
    ctx.arena_free(aid)
    assert [bool(int(x) & 1) for x in got[:, 0]] == [c["expected"] for c in fx["cases"]]


@pytest.mark.gpu
def test_gpu_arena64_bitsets_and_survivors_match_golden(ctx):
    from bloomsearch_amd.arena import entry_sets_from_strings, plan_blocks
    a = G["arena64"]
    blocks = [entry_sets_from_strings(f, t, ["g::" + x for x in t]) for f, t in a["blocks"]]
    plan = plan_blocks(blocks, a["fpr"])
    assert [[int(d["word_off"]), int(d["m"]), int(d["k"])] for d in plan.desc] == a["desc"]
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    assert words.astype("<u8").tobytes().hex() == a["words_hex"]
    cb = Q.compile_queries(a["expressions"])
    ops, poff, kinds = cb.arrays()
    terms = np.zeros(len(cb.term_strings), dtype=TERM_DTYPE)
    terms["h"] = ctx.hash_strings(cb.term_strings)
    terms["kind"] = kinds
    aid = ctx.arena_load(words, plan.desc)
    got = ctx.probe(aid, 64, terms, ops, poff)
    ctx.arena_free(aid)
    assert ["%016x" % int(x) for x in got[:, 0]] == a["survivors_hex"]
