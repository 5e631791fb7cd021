"""Seed sweep of bsg_build over filter sizes (GPU box):  python tools/fuzz_build.py [first_seed] [n_seeds]
Entry counts from 1 to a few million and false-positive rates from 0.5 to 1e-6 put m anywhere from a few bits to tens of
megabits: LDS-staged filters, filters just beyond LDS (global atomics by default), large ones (binned by 64 KiB window),
window counts of 1, 2, many, a last window of a few words; every eighth seed one bitset of 2^31 or 2^33 bits
(64-bit modulo, sliced global atomics); several filters per call; from entry bytes and from
precomputed hashes; with the binning threshold at its default and at zero.  Bitsets must equal the oracle's.
Exits non-zero on the first difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bloomsearch_amd._lib import DESC_DTYPE
from bloomsearch_amd.gpu import Context
from oracle import oracle as O


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    # one-entry context, and contexts of 2 / 3 entries that cut EVERY call into one part per entry (lab key 7: no minimum size),
    # so the sharded build (parts on threads, absolute indices over rebased staging, disjoint word ranges) is swept as well
    ctxs = [Context((0,)), Context((0, 0)), Context((0, 0, 0))]
    for c in ctxs[1:]:
        c.set_lab(7, 1)
    total = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        n_filters = int(rng.integers(1, 5))
        counts = [int(10 ** rng.uniform(0, 6.3)) for _ in range(n_filters)]
        if rng.random() < 0.3:
            counts[0] = 0
        fprs = [float(10 ** rng.uniform(-6, -0.3)) for _ in range(n_filters)]
        # entries: 8 random bytes + a counter (distinct), lengths 9..20
        blobs, lens = [], []
        for c in counts:
            ln = rng.integers(9, 21, size=c).astype(np.uint32)
            raw = rng.integers(0, 256, size=int(ln.sum()), dtype=np.uint8)
            blobs.append(raw); lens.append(ln)
        blob = np.concatenate(blobs) if blobs else np.zeros(0, dtype=np.uint8)
        ln = np.concatenate(lens) if lens else np.zeros(0, dtype=np.uint32)
        off = np.zeros(len(ln) + 1, dtype=np.uint32)
        np.cumsum(ln, out=off[1:])
        fstart = np.zeros(n_filters + 1, dtype=np.uint32)
        fstart[1:] = np.cumsum(counts)
        desc = np.zeros(n_filters, dtype=DESC_DTYPE)
        cursor = 0
        for f in range(n_filters):
            m, k = O.estimate_parameters(max(counts[f], 1), fprs[f])
            # every eighth seed gives one filter a geometry of 2^31 bits or more (the 64-bit-modulo instantiations): the caller
            # owns (m, k), the entries do not have to fill it
            if f == 0 and seed % 8 == 0:
                m = (1 << 31) + 12345 if seed % 16 else (1 << 33) + 7
                m += int(rng.integers(0, 1 << 20))
            desc[f] = (cursor, m, k, 0)
            cursor += ((m + 63) // 64 + 15) // 16 * 16
        n_words = max(cursor, 2)
        want = O.build_many(blob, off, fstart, desc.view(O.DESC_DTYPE), n_words)
        ctx = ctxs[seed % 3] if n_filters > 1 else ctxs[0]
        ctx.set_lab(6, 0 if seed % 2 else 4 << 20)
        got = ctx.build(blob, off, fstart, desc, n_words)
        if not np.array_equal(got, want):
            sys.exit("seed %d: bsg_build differs (counts %s, m %s, k %s)" % (seed, counts, desc["m"].tolist(), desc["k"].tolist()))
        hashed = ctx.build_hashed(ctx.hash_entries(blob, off), fstart, desc, n_words) if len(ln) else got
        if not np.array_equal(hashed, want):
            sys.exit("seed %d: bsg_build_hashed differs" % seed)
        total += int(sum(counts))
        if (seed - first) % 10 == 9:
            print("seed %d ok (%.1f M entries so far)" % (seed, total / 1e6), flush=True)
    for c in ctxs:
        c.close()
    print("done: %d calls, %.1f M entries, no difference" % (n, total / 1e6))


if __name__ == "__main__":
    main()
