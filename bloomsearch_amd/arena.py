"""Host-side planning of a block-filter arena (what flush.go:191-254 does per
partition buffer): size each block's three filters from its distinct entry
counts — buildSizedBloomFilter, ingest.go:139-140: (m, k) =
EstimateParameters(max(n, 1), fpr) — and lay them out in one word arena that
bsg_build fills and bsg_arena_load uploads.
"""
from __future__ import annotations

import numpy as np

from ._lib import DESC_DTYPE
from .gpu import estimate_parameters


class FilterPlan:
    """Packed entries grouped by filter + descriptors, ready for bsg_build."""

    def __init__(self, blob, off, fstart, desc, n_words, counts):
        self.blob, self.off, self.fstart, self.desc, self.n_words, self.counts = blob, off, fstart, desc, n_words, counts

    @property
    def n_blocks(self) -> int:
        return len(self.desc) // 3


def plan_blocks(blocks, fpr: float, absent=frozenset()) -> FilterPlan:
    """blocks: per block a list of 3 (u8 blob, u32 lengths) entry sets (field, token, field::token).
    absent: set of (block, kind) to leave nil (m == 0)."""
    blobs, lens = [], []
    fstart = [0]
    desc = np.zeros(len(blocks) * 3, dtype=DESC_DTYPE)
    counts = np.zeros((len(blocks), 3), dtype=np.int64)
    cursor = 0
    n_entries = 0
    cache: dict = {}
    for b, sets in enumerate(blocks):
        for c in range(3):
            blob, ln = sets[c]
            n = len(ln)
            counts[b, c] = n
            if (b, c) in absent:
                fstart.append(n_entries)
                continue
            key = max(n, 1)
            mk = cache.get(key)
            if mk is None:
                mk = cache[key] = estimate_parameters(key, fpr)
            m, k = mk
            d = desc[b * 3 + c]
            d["word_off"], d["m"], d["k"] = cursor, m, k
            cursor += ((m + 63) // 64 + 1) // 2 * 2
            blobs.append(np.asarray(blob, dtype=np.uint8))
            lens.append(np.asarray(ln, dtype=np.uint32))
            n_entries += n
            fstart.append(n_entries)
    blob = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
    ln = np.concatenate(lens) if lens else np.zeros(0, np.uint32)
    off = np.zeros(len(ln) + 1, dtype=np.uint64)
    np.cumsum(ln, out=off[1:])
    if off[-1] >= 2 ** 32:
        raise ValueError("entry blob exceeds the u32 offset range of one bsg_build call; split the plan")
    return FilterPlan(blob, off.astype(np.uint32), np.asarray(fstart, dtype=np.uint32), desc, max(cursor, 2), counts)


def entry_sets_from_strings(fields, tokens, field_tokens):
    """Convenience for tests: three iterables of str/bytes -> the (blob, lengths) triple."""
    out = []
    for s in (fields, tokens, field_tokens):
        bs = [x.encode() if isinstance(x, str) else bytes(x) for x in s]
        out.append((np.frombuffer(b"".join(bs), dtype=np.uint8), np.asarray([len(x) for x in bs], dtype=np.uint32)))
    return out
