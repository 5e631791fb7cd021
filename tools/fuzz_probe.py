"""Seed sweep of build + probe against the oracle (GPU box):  python tools/fuzz_probe.py [first_seed] [n_seeds]
Per seed: a few random arenas (block counts around the 64-block groups, false-positive rates from 1e-5 to 0.5, nil
filters, blocks without tokens), the device build compared with the oracle's bitsets, then random query batches — one
query, a handful, hundreds; few distinct terms (one-dispatch and few-term kernels) or thousands (many-term kernel) —
probed through every launch shape (group limit 1 / 3 / 32, fused, folded into one dispatch (k_probe_eval) or not, timed or not, sharded contexts) and through the one-call
bsg_query (strings in; kernel-argument fast path or the batch path inside the call), and as survivor rows (bsg_probe_many_rows), and compared
with the oracle's surviving-block sets bit for bit.  Exits non-zero on the first difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import helpers as H
from bloomsearch_amd import _lib, query as Q
from bloomsearch_amd._lib import DESC_DTYPE
from bloomsearch_amd.gpu import Context
from oracle import oracle as O


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctxs = {1: Context((0,)), 3: Context((0,) * 3)}
    n_cases = n_bits = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        fpr = float(10 ** rng.uniform(-5, -0.3))
        vocab_size = int(rng.choice([8, 40, 400, 5000]))
        plans = []
        for _ in range(int(rng.integers(1, 5))):
            nb = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, int(rng.integers(1, 400))]))
            plan, _, vocab = H.make_random_arena(rng, nb, fpr=fpr, absent_frac=float(rng.choice([0.0, 0.05, 0.5])),
                                                 max_tokens=int(rng.choice([5, 300, 3000])), vocab_size=vocab_size)
            plans.append(plan)
        nd = int(rng.choice([1, 1, 3]))
        ctx = ctxs[nd]
        arenas, all_words = [], []
        for plan in plans:
            words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
            if not np.array_equal(words, H.oracle_words(plan)):
                sys.exit("seed %d: device build differs from the oracle's bitsets" % seed)
            all_words.append(words)
            arenas.append(ctx.arena_load(words, plan.desc))
        for _ in range(3):
            nq = int(rng.choice([1, 1, 2, 3, 7, 12, 64, 256, 257, 600]))
            words_pool = vocab[: int(rng.choice([2, 6, len(vocab)]))]
            exprs_of_batch = [H.random_expression(rng, words_pool, None) if rng.random() > 0.03 else None for _ in range(nq)]
            cb = Q.compile_queries(exprs_of_batch)
            ops, poff, _ = cb.arrays()
            terms = H.gpu_terms(ctx, cb)
            bid = ctx.batch_create(terms, ops, poff)
            wants = [O.probe_batch(w, p.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff) for w, p in zip(all_words, plans)]
            order = [int(i) for i in rng.integers(0, len(arenas), size=int(rng.integers(1, 40)) if rng.random() > 0.1 else int(rng.integers(129, 400)))]
            ctx.set_probe_group(int(rng.choice([1, 3, 32, 0])))          # 0: the default — up to 1 024 arenas per dispatch, records in device memory beyond 128
            ctx.set_lab(11, int(rng.choice([0, 0, 1, 3, 8])))          # k_probe_eval: evaluation folded into the probe dispatch, 1 / 3 / 8 evaluators per tile
            flags = int(rng.choice([0, _lib.PROBE_NOFUSE, _lib.PROBE_TIMED]))
            got = ctx.probe_many([arenas[i] for i in order], bid, flags, cb.n_queries, [plans[i].n_blocks for i in order])
            for g, i in zip(got, order):
                if not np.array_equal(g, wants[i]):
                    sys.exit("seed %d: survivors differ (nq %d, %d terms, group order %s, flags %d, %d-entry context, arena %d)"
                             % (seed, nq, len(terms), order[:8], flags, nd, i))
                n_bits += g.size * 64
            # the same call as survivor ROWS (single-device contexts): whatever the tags, the rows expand to the same bitsets
            if nd == 1 and rng.random() < 0.5:
                from bloomsearch_amd.gpu import rows_to_dense
                NQ = cb.n_queries
                Gs = [(plans[i].n_blocks + 63) // 64 for i in order]
                rows = ctx.pinned_array(max(NQ * sum(Gs), 1) * 8).view(np.uint64)
                hdr = ctx.pinned_array(NQ * len(order) * 4).view(np.uint32)
                rows[:] = np.iinfo(np.uint64).max
                hdr[:] = np.iinfo(np.uint32).max
                # (every other time in the packed form where the arenas allow it: a run's payloads back to back, byte headers)
                packed = bool(rng.random() < 0.5) and max(Gs, default=0) <= 16
                ctx.probe_many_rows([arenas[i] for i in order], bid, rows, hdr, flags | (_lib.PROBE_ROWS_PACKED if packed else 0))
                o = 0
                for j, i in enumerate(order):
                    h = hdr.view(np.uint8)[j * NQ: (j + 1) * NQ] if packed else hdr[j * NQ: (j + 1) * NQ]
                    if not np.array_equal(rows_to_dense(h, rows[o: o + NQ * Gs[j]], plans[i].n_blocks, packed=packed), wants[i]):
                        sys.exit("seed %d: survivor rows differ (nq %d, %d terms, %d arenas, flags %d, arena %d, packed %s)" % (seed, nq, len(terms), len(order), flags, i, packed))
                    o += NQ * Gs[j]
                    n_bits += NQ * Gs[j] * 64
                ctx.pinned_free(rows.view(np.uint8))
                ctx.pinned_free(hdr.view(np.uint8))
            # the same batch through the one-call path (bsg_query: strings in, hashed on the host; one k_query_direct dispatch per
            # device when <= 16 terms / 128 program words / 32 arenas fit the kernel arguments, the batch path inside the call otherwise)
            sub = [int(i) for i in rng.integers(0, len(arenas), size=int(rng.choice([1, 2, 5, 33])))]
            gq = ctx.query([arenas[i] for i in sub], [plans[i].n_blocks for i in sub], cb)
            for g, i in zip(gq, sub):
                if not np.array_equal(g, wants[i]):
                    sys.exit("seed %d: bsg_query differs (nq %d, %d terms, %d arenas, %d-entry context, arena %d)" % (seed, nq, len(terms), len(sub), nd, i))
                n_bits += g.size * 64
            # ... and from several threads at once, every thread its own handful of the batch's queries over its own arena subset, the
            # collector told to wait for company (lab key 15): calls are merged — hot arenas as one batch, the rest as jobs of one
            # dispatch — and every call must still get exactly its rows
            if cb.n_queries >= 2 and rng.random() < 0.5:
                import threading
                ctx.set_lab(15, (3000 << 16) | 5)
                ctx.set_lab(16, int(rng.choice([0, 2, 24])))
                errs = []

                def one(t, seed2):
                    r2 = np.random.default_rng(seed2)
                    try:
                        for _ in range(3):
                            qs = [int(x) for x in r2.integers(0, cb.n_queries, size=int(r2.choice([1, 1, 2, 4])))]
                            sub2 = [int(i) for i in r2.integers(0, len(arenas), size=int(r2.choice([1, 1, 2, 5])))]
                            small = Q.compile_queries([exprs_of_batch[q] for q in qs])
                            g2 = ctx.query([arenas[i] for i in sub2], [plans[i].n_blocks for i in sub2], small)
                            for g, i in zip(g2, sub2):
                                if not np.array_equal(g, wants[i][qs]):
                                    errs.append("thread %d: queries %s on arena %d" % (t, qs, i))
                    except BaseException as exc:  # noqa: BLE001
                        errs.append(repr(exc))
                ths = [threading.Thread(target=one, args=(t, seed * 100 + t)) for t in range(6)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                ctx.set_lab(15, 0)
                ctx.set_lab(16, 8)
                if errs:
                    sys.exit("seed %d: concurrent bsg_query differs (%d-entry context): %s" % (seed, nd, errs[:3]))
                n_bits += 6 * 3 * 64
            n_cases += 1
            ctx.batch_free(bid)
        ctx.set_probe_group(0)
        ctx.set_lab(11, 0)
        ctx.timing_read(reset=True)
        for a in arenas:
            ctx.arena_free(a)
        if (seed - first) % 10 == 9:
            print("seed %d ok (%d batches, %.1f M (query, block) verdicts so far)" % (seed, n_cases, n_bits / 1e6), flush=True)
    for c in ctxs.values():
        c.close()
    print("done: %d batches, %.1f M verdicts, no difference" % (n_cases, n_bits / 1e6))


if __name__ == "__main__":
    main()
