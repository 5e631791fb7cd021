"""Worker of test_or_allreduce_loopback_gpu.py (run as a script in a process of its own: BSG_RCCL_LIBRARY is read once per
process).  Runs bsg_or_allreduce at world sizes > 1 on ONE GPU through tests/loopback_ccl.cpp — ranks are threads, or the
entries of one multi-entry context — and compares every rank's result with numpy's OR over every rank's filters."""
import sys
import threading

import numpy as np

from bloomsearch_amd.gpu import Context
from bloomsearch_amd._lib import DESC_DTYPE


def make_arena(rng, n_blocks, m, k):
    nw = (m + 63) // 64
    stride = (nw + 15) // 16 * 16
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    words = np.zeros(max(n_blocks * stride, 2), dtype=np.uint64)
    for b in range(n_blocks):
        desc[b * 3 + 1] = (b * stride, m, k, 0)
        w = rng.integers(0, 1 << 63, nw, dtype=np.uint64) & rng.integers(0, 1 << 63, nw, dtype=np.uint64) \
            & rng.integers(0, 1 << 63, nw, dtype=np.uint64)                      # about one bit in eight
        w[rng.integers(0, nw)] |= np.uint64(1) << np.uint64(63 if (m & 63) == 0 else 0)
        if m & 63:
            w[-1] &= np.uint64((1 << (m & 63)) - 1)
        words[b * stride: b * stride + nw] = w
    return words, desc, nw


def or_of(words, desc, nw):
    out = np.zeros(nw, dtype=np.uint64)
    for b in range(len(desc) // 3):
        o = int(desc[b * 3 + 1]["word_off"])
        out |= words[o: o + nw]
    return out


def ranks_as_threads(world, nw_target, rounds=3):
    m = nw_target * 64 - 13
    k = 7
    rng = np.random.default_rng(1000 + world * 7 + nw_target)
    uid = Context.comm_unique_id()
    arenas = [[make_arena(rng, int(rng.integers(0 if r else 1, 6)), m, k) for r in range(world)] for _ in range(rounds)]
    wants = []
    for rnd in range(rounds):
        w = np.zeros(nw_target, dtype=np.uint64)
        for words, desc, nw in arenas[rnd]:
            w |= or_of(words, desc, nw)
        wants.append(w)
    errors = []
    barrier = threading.Barrier(world)

    def rank(r):
        try:
            with Context((0,)) as ctx:
                ctx.comm_init(uid, r, world)
                for rnd in range(rounds):
                    words, desc, nw = arenas[rnd][r]
                    aid = ctx.arena_load(words, desc)
                    got = ctx.or_allreduce(aid, 1, nw)
                    if not np.array_equal(got, wants[rnd]):
                        errors.append("world %d rank %d round %d: %d words differ" % (world, r, rnd, int((got != wants[rnd]).sum())))
                    ctx.arena_free(aid)
                barrier.wait(timeout=60)
                ctx.comm_destroy()
        except Exception as e:                                  # noqa: BLE001 — reported by the parent
            errors.append("world %d rank %d: %r" % (world, r, e))
            barrier.abort()

    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    if any(t.is_alive() for t in th):
        errors.append("world %d: a rank is still waiting after 120 s" % world)
    return errors


def entries_of_one_context(entries, n_blocks, nw_target):
    m = nw_target * 64 - (0 if nw_target % 2 else 21)
    rng = np.random.default_rng(77 + entries + n_blocks)
    words, desc, nw = make_arena(rng, n_blocks, m, 9)
    want = or_of(words, desc, nw)
    with Context((0,) * entries) as ctx:
        ctx.comm_init(None)
        aid = ctx.arena_load(words, desc)
        got = ctx.or_allreduce(aid, 1, nw)
        same_as_local = np.array_equal(ctx.or_reduce(aid, 1, nw), want)
        ctx.arena_free(aid)
        ctx.comm_destroy()
    out = []
    if not np.array_equal(got, want):
        out.append("context of %d entries, %d blocks, %d words: %d words differ" % (entries, n_blocks, nw, int((got != want).sum())))
    if not same_as_local:
        out.append("bsg_or_reduce differs on %d entries" % entries)
    return out


def c5_at_full_block_count(world=8, n_blocks=10000, rows=5):
    """BASELINE configs[4] at its stated size: 10 000 FIXED-GEOMETRY token filters (m, k) = EstimateParameters(|union|, p),
    block b on rank b % 8 (1 250 per rank), every rank ORs its shard (k_or_reduce_blocks) and the partials go through the
    library's slice exchange + k_or_words + all-gather.  Every rank's result must equal the ORACLE's build of the union of
    all 10 000 blocks' token sets at that geometry (OR == rebuild exactly under a fixed geometry)."""
    from bloomsearch_amd.gpu import pack_entries
    from oracle import oracle as O
    from tests.test_configs_gpu import gen_blocks
    blocks = gen_blocks(np.arange(n_blocks), rows)
    per_block, union = [], set()
    for sets in blocks:
        blob, lens = sets[1]
        off = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        raw = blob.tobytes()
        toks = [raw[off[i]: off[i + 1]] for i in range(len(lens))]
        per_block.append(toks)
        union.update(toks)
    m, k = O.estimate_parameters(len(union), 0.001)
    nw = O.words_for(m)
    stride = (nw + 15) // 16 * 16
    want = O.Filter(m, k)
    for t in union:
        want.add(t)
    uid = Context.comm_unique_id()
    errors = []
    barrier = threading.Barrier(world)

    def rank(r):
        try:
            mine = list(range(r, n_blocks, world))
            desc = np.zeros(len(mine) * 3, dtype=DESC_DTYPE)
            fstart, ents = [0], []
            for i, b in enumerate(mine):
                fstart.append(len(ents))                 # field: absent
                desc[i * 3 + 1] = (i * stride, m, k, 0)
                ents += per_block[b]
                fstart += [len(ents), len(ents)]         # field::token: absent
            blob, off = pack_entries(ents)
            with Context((0,)) as ctx:
                words = ctx.build(blob, off, np.asarray(fstart, dtype=np.uint32), desc, len(mine) * stride)
                ctx.comm_init(uid, r, world)
                seen, me, from_lib = ctx.comm_info()
                if (seen, me) != (world, r) or not from_lib:
                    errors.append("rank %d: the communicator reports rank %d of %d" % (r, me, seen))
                aid = ctx.arena_load(words, desc)
                del words
                got = ctx.or_allreduce(aid, 1, nw)
                if not np.array_equal(got, want.words):
                    errors.append("c5 world %d rank %d: %d words differ from the oracle's build of the union" % (world, r, int((got != want.words).sum())))
                ctx.arena_free(aid)
                barrier.wait(timeout=120)
                ctx.comm_destroy()
        except Exception as e:                                  # noqa: BLE001 — reported by the parent
            errors.append("c5 world %d rank %d: %r" % (world, r, e))
            barrier.abort()

    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    if any(t.is_alive() for t in th):
        errors.append("c5 world %d: a rank is still waiting" % world)
    print("c5: %d fixed-geometry filters (m = %d, k = %d, union of %d tokens), %d per rank on %d ranks: %s"
          % (n_blocks, m, k, len(union), n_blocks // world, world, "ok" if not errors else "FAILED"))
    return errors


def main():
    if "c5" in sys.argv[1:]:
        errors = c5_at_full_block_count()
        for e in errors:
            print("FAIL", e)
        sys.exit(1 if errors else 0)
    errors = []
    for world, nw in ((2, 1001), (3, 1001), (4, 1001), (4, 5), (4, 3), (8, 4099), (2, 1)):
        e = ranks_as_threads(world, nw)
        print("ranks as threads: world %d, %d words, 3 rounds: %s" % (world, nw, "ok" if not e else "FAILED"))
        errors += e
    for entries, n_blocks, nw in ((2, 7, 1001), (3, 2, 640), (3, 10, 1001), (8, 29, 333), (4, 4, 2)):
        e = entries_of_one_context(entries, n_blocks, nw)
        print("one context: %d entries, %d blocks, %d words: %s" % (entries, n_blocks, nw, "ok" if not e else "FAILED"))
        errors += e
    for e in errors:
        print("FAIL", e)
    print("loopback or_allreduce: %s" % ("ok" if not errors else "%d failures" % len(errors)))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
