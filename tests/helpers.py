"""Shared test helpers: random arenas + random query trees, and the oracle-side
evaluation they are compared with.  The oracle is only ever the checker here."""
from __future__ import annotations

import numpy as np

from bloomsearch_amd import query as Q
from bloomsearch_amd._lib import DESC_DTYPE, TERM_DTYPE
from bloomsearch_amd.arena import entry_sets_from_strings, plan_blocks
from oracle import oracle as O


def device_ids(n: int):
    """Device ids for a context of n entries: REAL devices 0 .. n-1 where the box has them (peer copies, bsg_peer_access, per-device
    PCIe slices then run between distinct GPUs), else n aliases of device 0 — all a 1-GPU box can offer ("multi-device
    contexts are only ever (0,)*N aliases").  BSG_TEST_ALIAS_DEVICES=1 forces the aliases.  Asks the library, not torch: torch brings
    its own copy of the HIP / HSA runtime into the process, after which /opt/rocm's librccl — which the library binds for
    bsg_comm_init — finds an HSA runtime nobody initialised ("no ROCm-capable device is detected")."""
    import os
    from bloomsearch_amd import _lib
    if os.environ.get("BSG_TEST_ALIAS_DEVICES") != "1" and _lib.load().bsg_device_count() >= n:
        return tuple(range(n))
    return (0,) * n


def device_free_bytes() -> int:
    """hipMemGetInfo's free bytes of the current device, through the HIP runtime the library itself is linked against (not torch's)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    rc = hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    assert rc == 0, rc
    return int(free.value)


def random_block_strings(rng, n_fields, n_tokens, vocab):
    fields = ["f%d" % i for i in rng.choice(40, size=min(n_fields, 40), replace=False)]
    toks = sorted({vocab[i] for i in rng.integers(0, len(vocab), size=n_tokens)})
    fts = sorted({fields[rng.integers(0, len(fields))] + "::" + t for t in toks}) if fields else []
    return fields, toks, fts


def make_random_arena(rng, n_blocks, fpr=0.01, max_tokens=3000, absent_frac=0.05, vocab_size=5000):
    vocab = ["tok%d" % i for i in range(vocab_size)] + ["Ünï%d" % i for i in range(50)]
    blocks_str, blocks = [], []
    absent = set()
    for b in range(n_blocks):
        n_tokens = int(rng.integers(0, max_tokens)) if rng.random() > 0.1 else 0
        f, t, ft = random_block_strings(rng, int(rng.integers(1, 12)), n_tokens, vocab)
        blocks_str.append((f, t, ft))
        blocks.append(entry_sets_from_strings(f, t, ft))
        for c in range(3):
            if rng.random() < absent_frac:
                absent.add((b, c))
    plan = plan_blocks(blocks, fpr, absent)
    return plan, blocks_str, vocab


def oracle_words(plan):
    """The arena's words as the oracle builds them from the same packed entries."""
    return O.build_many(plan.blob, plan.off, plan.fstart, plan.desc.view(O.DESC_DTYPE), plan.n_words)


def random_expression(rng, vocab, fields, depth=0):
    r = rng.random()
    if depth >= 3 or r < 0.45:
        kind = rng.integers(0, 3)
        tok = vocab[rng.integers(0, len(vocab))] if rng.random() < 0.8 else "absent%d" % rng.integers(0, 1000)
        fld = "f%d" % rng.integers(0, 45)
        if kind == 0:
            return Q.Field(fld)
        if kind == 1:
            return Q.Token(tok)
        return Q.FieldToken(fld, tok)
    if r < 0.50:
        return {"ExpressionType": "CONDITION", "Condition": None}          # nil condition => true
    if r < 0.53:
        return {"ExpressionType": "XOR", "Children": []}                   # unknown expression => false
    if r < 0.56:
        return {"ExpressionType": "CONDITION", "Condition": {"Type": "BOGUS", "Field": "x", "Token": "y"}}
    n = int(rng.integers(0, 5))
    kids = [random_expression(rng, vocab, fields, depth + 1) for _ in range(n)]
    return Q.And(*kids) if rng.random() < 0.5 else Q.Or(*kids)


def oracle_terms(cb):
    """bsg_term table via the oracle's base hashes (checker side)."""
    terms = np.zeros(len(cb.term_strings), dtype=TERM_DTYPE)
    for i, s in enumerate(cb.term_strings):
        terms["h"][i] = O.base_hashes(s)
        terms["kind"][i] = cb.term_kinds[i]
    return terms


def gpu_terms(ctx, cb):
    """bsg_term table via bsg_hash_entries (product side)."""
    terms = np.zeros(len(cb.term_strings), dtype=TERM_DTYPE)
    if len(cb.term_strings):
        terms["h"] = ctx.hash_strings(cb.term_strings)
        terms["kind"] = np.asarray(cb.term_kinds, dtype=np.uint32)
    return terms
