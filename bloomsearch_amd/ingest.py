"""Device ingest glue (test / bench): rows -> bsg_ingest_* -> filters, with the host walker finishing
the rows the device walker hands back (include/bloomgpu.h "device ingest").

Mirrors the flush worker's per-partition loop (flush.go:179-254): one set per partition buffer,
one parent per file, (m, k) sized on the host from the exact distinct counts (buildFilters,
ingest.go:127-145: n' = max(count, 1)).
"""
from __future__ import annotations

import numpy as np

from . import host
from ._lib import DESC_DTYPE
from .gpu import Context, estimate_parameters


class IngestResult:
    def __init__(self, counts, status, desc, words, stats, fallback_rows):
        self.counts, self.status, self.desc, self.words, self.stats, self.fallback_rows = counts, status, desc, words, stats, fallback_rows

    def filter_words(self, set_index: int, kind: int) -> np.ndarray:
        d = self.desc[set_index * 3 + kind]
        nw = (int(d["m"]) + 63) // 64
        return self.words[int(d["word_off"]): int(d["word_off"]) + nw]


def plan_desc(counts: np.ndarray, fpr: float):
    """counts [n, 3] -> (desc [n*3], n_words): right-sized geometry, filters on 16-byte boundaries."""
    desc = np.zeros(counts.shape[0] * 3, dtype=DESC_DTYPE)
    cursor = 0
    cache: dict = {}
    for i, n in enumerate(counts.reshape(-1)):
        key = max(int(n), 1)
        mk = cache.get(key)
        if mk is None:
            mk = cache[key] = estimate_parameters(key, fpr)
        desc[i]["word_off"], desc[i]["m"], desc[i]["k"] = cursor, mk[0], mk[1]
        cursor += ((mk[0] + 63) // 64 + 1) // 2 * 2
    return desc, max(cursor, 2)


def host_walk_entries(rows, row_ids, set_of_row):
    """The host walker (walker.hpp through bsh_entry_sets_*) over the fallback rows ->
    (entries, set_of_entry, kind_of_entry)."""
    by_set: dict = {}
    for r in row_ids:
        by_set.setdefault(int(set_of_row[r]), []).append(rows[int(r)])
    entries, sets, kinds = [], [], []
    for s, rs in by_set.items():
        es = host.EntrySets()
        for row in rs:
            try:
                es.index_row(row)
            except host.HostError:
                pass  # malformed row: what the walker saw before the error stays (entry_sets.hpp)
        for kind in range(3):
            blob, ln = es.export(kind)
            off = np.concatenate([[0], np.cumsum(ln, dtype=np.int64)])
            raw = blob.tobytes()
            for i in range(len(ln)):
                entries.append(raw[off[i]: off[i + 1]])
                sets.append(s)
                kinds.append(kind)
    return entries, sets, kinds


def device_ingest(ctx: Context, row_sets, fpr: float, parent_of_set=None, n_parents: int = 0, slots_hint=None,
                  keep: bool = False, flags: int = 0) -> IngestResult:
    """row_sets: list (one per set) of lists of row bytes."""
    rows = [r for rs in row_sets for r in rs]
    first = np.zeros(len(row_sets) + 1, dtype=np.uint32)
    first[1:] = np.cumsum([len(rs) for rs in row_sets])
    ing = ctx.ingest_rows(rows, first, parent_of_set, n_parents, slots_hint, flags)
    try:
        fb = ctx.ingest_fallback_rows(ing)
        if len(fb):
            set_of_row = np.repeat(np.arange(len(row_sets)), np.diff(first.astype(np.int64)))
            entries, sets, kinds = host_walk_entries(rows, fb, set_of_row)
            if entries:
                ctx.ingest_add_entries(ing, entries, sets, kinds)
        n_total = len(row_sets) + n_parents
        counts, status = ctx.ingest_finish(ing, n_total)
        desc, n_words = plan_desc(counts, fpr)
        words = ctx.ingest_build(ing, desc, n_words)
        stats = ctx.ingest_stats(ing)
        return IngestResult(counts, status, desc, words, stats, fb)
    finally:
        if not keep:
            ctx.ingest_free(ing)
