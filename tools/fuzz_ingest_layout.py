"""Seed sweep over the LAYOUT of a device ingest (GPU box):  python tools/fuzz_ingest_layout.py [first_seed] [n_seeds]
The rows are the synthetic log rows; what varies is everything around them: the number of sets and their sizes (empty sets,
one-row sets, sets that end inside a 64-row tile or inside a wave's run of rows), which parent a set feeds (none, one of
several), the upload chunk size (from 64 KiB — dozens of chunks, launches of a few hundred rows — to one chunk), table size
hints that force the tables to grow and the walk to be repeated, validated or trusted, the number of entries of the context
(1, 2, 3, 5: parts, partial parents merged across parts) and the route of the file-level union (LDS partitions from too coarse to
far too fine, global tables).  Counts and bitsets of every set and
every parent must equal the oracle's build of the oracle's sets.  Exits non-zero on the first difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bloomsearch_amd import ingest as I, synth
from bloomsearch_amd.gpu import Context
from oracle import oracle as O
from oracle import walker_oracle as W

FPR = 0.01


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    # contexts of 1, 2, 3 and 5 entries (all on device 0): the multi-entry ones cut every ingest into parts whatever its size
    # (lab key 8), so partial parents are packed, moved and merged (merge_parents) and every part builds its own sets
    ctxs = [Context((0,) * k) for k in (1, 2, 3, 5)]
    for c in ctxs[1:]:
        c.set_lab(8, 1)
    pool = synth.rows_json(0, 6000)
    entries = [W.index_row(r) for r in pool]                 # per row (fields, tokens, field_tokens)
    n_rows = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        n_sets = int(rng.integers(1, 12))
        sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, 200, int(rng.integers(0, 900))])) for _ in range(n_sets)]
        start = int(rng.integers(0, len(pool) - sum(sizes))) if sum(sizes) < len(pool) else 0
        row_sets, idx = [], start
        for s in sizes:
            row_sets.append(pool[idx: idx + s])
            idx += s
        n_parents = int(rng.integers(0, 4))
        parent_of = [int(rng.integers(0, n_parents)) if n_parents and rng.random() < 0.8 else 0xFFFFFFFF for _ in range(n_sets)]
        hint = None
        if rng.random() < 0.3:
            hint = [int(rng.choice([64, 256, 1024])) for _ in range(n_sets * 3)]
        ctx = ctxs[int(rng.integers(0, len(ctxs)))]
        ctx.set_ingest_chunk(int(rng.choice([1 << 16, 1 << 17, 1 << 20, 64 << 20])))
        # the file-level union: LDS partitions (default), global tables, too coarse a start (retry / fallback), finer partitions
        ctx.set_lab(9, int(rng.choice([0, 0, 1])))
        ctx.set_lab(10, int(rng.choice([0, 0, 2, 4, 14, 33, 36, 42])))
        res = I.device_ingest(ctx, row_sets, FPR, parent_of_set=parent_of if n_parents else None, n_parents=n_parents, slots_hint=hint,
                              flags=int(rng.integers(0, 2)))
        want = []
        base = start
        for s in sizes:
            sets = (set(), set(), set())
            for e in entries[base: base + s]:
                for k in range(3):
                    sets[k].update(e[k])
            want.append(sets)
            base += s
        for p in range(n_parents):
            sets = (set(), set(), set())
            for s_, par in enumerate(parent_of):
                if par == p:
                    for k in range(3):
                        sets[k].update(want[s_][k])
            want.append(sets)
        for i, sets in enumerate(want):
            for kind in range(3):
                if int(res.counts[i, kind]) != len(sets[kind]):
                    sys.exit("seed %d set %d kind %d: device counts %d, oracle %d (sizes %s parents %s)"
                             % (seed, i, kind, int(res.counts[i, kind]), len(sets[kind]), sizes, parent_of))
                f = O.build_sized(sorted(sets[kind]), FPR)
                d = res.desc[i * 3 + kind]
                if (int(d["m"]), int(d["k"])) != (f.m, f.k) or not np.array_equal(res.filter_words(i, kind), f.words):
                    sys.exit("seed %d set %d kind %d: bitset differs" % (seed, i, kind))
        if len(res.fallback_rows):
            sys.exit("seed %d: %d synthetic rows handed to the host" % (seed, len(res.fallback_rows)))
        n_rows += sum(sizes)
        if (seed - first) % 20 == 19:
            print("seed %d ok (%d rows so far)" % (seed, n_rows), flush=True)
    for c in ctxs:
        c.close()
    print("done: %d layouts, %d rows, no difference" % (n, n_rows))


if __name__ == "__main__":
    main()
