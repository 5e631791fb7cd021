"""BASELINE configs[3] (C4) and configs[4] (C5) on the GPU against the oracle, and the launch-grouping machinery they
ride on: one dispatch per group of arenas, survivors leaving on a copy stream, multi-device contexts interleaving
their shards on the host.  Block COUNTS are the configs' (1 250 per GPU at 8-way, 10 000 in total); rows per block are
kept small so the CPU side of the test (generator + oracle) runs in seconds."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from bloomsearch_amd import _lib, query as Q, synth
from bloomsearch_amd._lib import DESC_DTYPE, BloomGpuError
from bloomsearch_amd.arena import plan_blocks
from bloomsearch_amd.gpu import Context, pack_entries
from oracle import oracle as O
from tests import helpers as H
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu

ROWS = 5              # rows per block in these tests: few enough that a block lacks most low-cardinality values, so an Or of 8 leaves prunes


def _gen(args):
    b, rows = args
    return synth.block_entry_sets(b * rows, rows)


def gen_blocks(ids, rows=ROWS):
    jobs = [(int(b), rows) for b in ids]
    if len(jobs) < 64:
        return [_gen(j) for j in jobs]
    with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
        return pool.map(_gen, jobs, chunksize=64)


class HipMem:
    """Raw device memory for the test (hipMalloc through ctypes on the HIP runtime the library itself links)."""

    def __init__(self, n_bytes):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(n_bytes)) == 0
        self.n = n_bytes

    def read(self):
        out = np.zeros(self.n // 8, dtype=np.uint64)
        assert self.hip.hipMemcpy(self.C.c_void_p(out.ctypes.data), self.ptr, self.C.c_size_t(self.n), 2) == 0   # hipMemcpyDeviceToHost
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def c4_batch(ctx, nq, seed=99):
    cb = Q.compile_queries(synth.make_queries(nq, "c4", seed))
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    return cb, terms, ops, poff


def test_c4_query_shape_over_a_1250_block_shard(ctx):
    """configs[3] as one of 8 ranks sees it: blocks b = 8 j + 3 of a 10 000-block set, Q 8-term Or(FieldToken) queries."""
    ids = np.arange(3, 10000, 8)
    assert len(ids) == 1250
    plan = plan_blocks(gen_blocks(ids), 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    assert np.array_equal(words, H.oracle_words(plan))
    cb, terms, ops, poff = c4_batch(ctx, 64)
    aid = ctx.arena_load(words, plan.desc)
    got = ctx.probe(aid, 1250, terms, ops, poff)
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(got, want)
    # the Or of 8 leaves must actually prune somewhere and keep somewhere, or the test shows nothing
    ones = sum(bin(int(x)).count("1") for x in got.ravel())
    assert 0 < ones < 64 * 1250
    ctx.arena_free(aid)


def test_c4_10000_blocks_on_an_8_entry_multi_device_context():
    """The whole configs[3] set behind ONE context that lists 8 devices (device 0 eight times: eight streams, eight
    shards, block b on entry b % 8) as 10 files of 1 000 blocks probed by one bsg_probe_many; the host-interleaved
    survivors must equal the oracle's over every file, and a single-device context's."""
    n_files, per_file, nq = 10, 1000, 48
    with Context(device_ids(8)) as mctx, Context((0,)) as sctx:
        cb, terms, ops, poff = c4_batch(sctx, nq, seed=7)
        mb = mctx.batch_create(terms, ops, poff)
        sb = sctx.batch_create(terms, ops, poff)
        m_ids, s_ids, wants = [], [], []
        for f in range(n_files):
            plan = plan_blocks(gen_blocks(range(f * per_file, (f + 1) * per_file)), 0.001)
            words = sctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
            m_ids.append(mctx.arena_load(words, plan.desc))
            s_ids.append(sctx.arena_load(words, plan.desc))
            wants.append(O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff))
        got_m = mctx.probe_many(m_ids, mb, 0, nq, [per_file] * n_files)
        got_s = sctx.probe_many(s_ids, sb, 0, nq, [per_file] * n_files)
        for f in range(n_files):
            assert np.array_equal(got_m[f], wants[f]), f
            assert np.array_equal(got_s[f], wants[f]), f
        # one arena through bsg_probe_batch on the sharded context: same interleave
        assert np.array_equal(mctx.probe_batch(m_ids[3], mb, nq, per_file), wants[3])


@pytest.mark.parametrize("n_entries", [2, 5, 7])
def test_sharded_context_interleaves_any_block_count(n_entries):
    """One context over n entries (block b on entry b % n): the host interleave of the shards' survivor bitsets (bit-field
    extract + parallel deposit per output word) for block counts around the word and shard boundaries, dense and sparse
    survivors, against the oracle."""
    rng = np.random.default_rng(1000 + n_entries)
    with Context(device_ids(n_entries)) as mctx:
        for n_blocks in (1, n_entries - 1, n_entries, n_entries + 1, 63, 64, 65, 64 * n_entries - 1, 64 * n_entries + 1, 321):
            plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.05, max_tokens=200, vocab_size=30)
            words = mctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
            cb = Q.compile_queries([None, Q.Token("absent")] + [H.random_expression(rng, vocab, None) for _ in range(70)])
            ops, poff, _ = cb.arrays()
            terms = H.gpu_terms(mctx, cb)
            aid = mctx.arena_load(words, plan.desc)
            want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
            assert np.array_equal(mctx.probe(aid, n_blocks, terms, ops, poff), want), n_blocks
            # a single interactive query: every entry answers its shard in one dispatch (k_probe_direct), same interleave
            c1 = Q.compile_queries([Q.And(Q.Token(vocab[0]), Q.Or(Q.Token(vocab[1]), Q.Field("f3")))])
            o1, p1, _ = c1.arrays()
            t1 = H.gpu_terms(mctx, c1)
            w1 = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), t1.view(O.TERM_DTYPE), o1, p1)
            assert np.array_equal(mctx.probe(aid, n_blocks, t1, o1, p1), w1), n_blocks
            mctx.arena_free(aid)


def test_c5_or_reduce_of_1250_fixed_geometry_blocks_equals_oracle_build_of_the_union(ctx):
    """configs[4] as one rank sees it: 1 250 token filters built at the FILE-level geometry (m, k) =
    EstimateParameters(n_union, p); their OR must equal the oracle's build of the union's entries at that geometry
    (the only condition under which OR == rebuild)."""
    n_blocks = 1250
    blocks = gen_blocks(np.arange(5, 10000, 8))
    union = set()
    per_block = []
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
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    fstart, ents = [0], []
    for b in range(n_blocks):
        fstart.append(len(ents))                 # field: absent
        desc[b * 3 + 1] = (b * stride, m, k, 0)
        ents += per_block[b]
        fstart += [len(ents), len(ents)]         # field::token: absent
    blob, off = pack_entries(ents)
    words = ctx.build(blob, off, np.asarray(fstart, dtype=np.uint32), desc, n_blocks * stride)
    aid = ctx.arena_load(words, desc)
    got = ctx.or_reduce(aid, 1, nw)
    ctx.arena_free(aid)
    want = O.Filter(m, k)
    for t in union:
        want.add(t)
    assert np.array_equal(got, want.words)
    # and on a context that shards the same arena over 8 entries (partials combined inside the library)
    with Context(device_ids(8)) as mctx:
        aid = mctx.arena_load(words, desc)
        assert np.array_equal(mctx.or_reduce(aid, 1, nw), want.words)


def test_rccl_or_allreduce_inside_the_library_world_of_one(ctx):
    """bsg_comm_init / bsg_or_allreduce: librccl bound at run time, a communicator of one rank on this box's single GPU
    (the N > 1 exchange runs in bench.py under torchrun): all-gather + OR must reproduce bsg_or_reduce, and the
    in-place device form must leave the partial unchanged."""
    from bloomsearch_amd.gpu import Context as Cx
    rng = np.random.default_rng(41)
    universe = ["tok%d" % i for i in range(5000)]
    m, k = O.estimate_parameters(len(universe), 0.001)
    nw = O.words_for(m)
    stride = (nw + 15) // 16 * 16
    n_blocks = 9
    desc = np.zeros(n_blocks * 3, dtype=DESC_DTYPE)
    fstart, ents = [0], []
    for b in range(n_blocks):
        fstart.append(len(ents))
        desc[b * 3 + 1] = (b * stride, m, k, 0)
        ents += [t for t in universe if rng.random() < 0.2]
        fstart += [len(ents), len(ents)]
    blob, off = pack_entries(ents)
    words = ctx.build(blob, off, np.asarray(fstart, dtype=np.uint32), desc, n_blocks * stride)
    aid = ctx.arena_load(words, desc)
    want = ctx.or_reduce(aid, 1, nw)
    with pytest.raises(BloomGpuError):
        ctx.or_allreduce(aid, 1, nw)                        # no communicator yet
    ctx.comm_init(Cx.comm_unique_id(), 0, 1)
    assert np.array_equal(ctx.or_allreduce(aid, 1, nw), want)
    dmem = HipMem(nw * 8)
    assert dmem.hip.hipMemcpy(dmem.ptr, dmem.C.c_void_p(want.ctypes.data), dmem.C.c_size_t(nw * 8), 1) == 0   # HostToDevice
    ctx.or_allreduce_dev([dmem.ptr.value], nw)
    assert np.array_equal(dmem.read(), want)
    dmem.free()
    ctx.comm_destroy()
    ctx.arena_free(aid)


def test_grouped_launches_equal_one_launch_per_arena(ctx):
    """A dispatch covers a GROUP of arenas (per-arena pointers in the kernel arguments): any grouping — one arena per
    launch, 3, 32, 64, more arenas than one group holds, arenas of different sizes, an arena without blocks, fused or not —
    must return what one probe per arena returns."""
    rng = np.random.default_rng(5)
    plans, vocab = [], None
    for n_blocks in (70, 3, 129, 64, 200, 1, 65):
        plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.02)
        plans.append((plan, ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)))
    # two batches: thousands of distinct terms (k_probe_terms_many, never fused) and a few dozen (k_probe_terms / k_probe_fused)
    batches = []
    for n_queries, words in ((520, vocab), (60, vocab[:12])):
        cb = Q.compile_queries([None] + [H.random_expression(rng, words, None) for _ in range(n_queries)])
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        batches.append((cb, ops, poff, terms, ctx.batch_create(terms, ops, poff)))
    assert np.bincount(batches[0][3]["kind"]).max() > 128 and np.bincount(batches[1][3]["kind"]).max() <= 128
    arenas, nbs = [], []
    for plan, words in plans:
        arenas.append(ctx.arena_load(words, plan.desc))
        nbs.append(plan.n_blocks)
    arenas.append(ctx.arena_load(np.zeros(2, dtype=np.uint64), np.zeros(0, dtype=DESC_DTYPE))); nbs.append(0)
    order = [int(i) for i in rng.integers(0, len(arenas), size=75)] + [7, 0, 7]
    try:
        for cb, ops, poff, terms, bid in batches:
            wants = [O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff) for plan, words in plans]
            wants.append(np.zeros((cb.n_queries, 0), dtype=np.uint64))
            for limit in (1, 3, 32, 64):
                ctx.set_probe_group(limit)
                for fold in (0, 8, 2):                      # lab key 11: k_probe_eval (evaluation folded into the probe dispatch) with 8 / 2 evaluators per tile
                    ctx.set_lab(11, fold)
                    for flags in (0, _lib.PROBE_NOFUSE, _lib.PROBE_TIMED):
                        got = ctx.probe_many([arenas[i] for i in order], bid, flags, cb.n_queries, [nbs[i] for i in order])
                        for g, i in zip(got, order):
                            assert np.array_equal(g, wants[i]), (len(terms), limit, fold, flags, i)
    finally:
        ctx.set_probe_group(0)
        ctx.set_lab(11, 0)
    t = ctx.timing_read()
    # few-term batches: k_probe_fused for small groups, k_probe_eval when folding is on; the many-term batch and every NOFUSE call: the two kernels
    assert t.n_fused > 0 and t.n_folded > 0 and t.n_probes > 0 and t.n_eval > 0 and t.n_fused_arenas >= t.n_fused and t.n_folded_arenas >= t.n_folded
    for a in arenas:
        ctx.arena_free(a)
    for b in batches:
        ctx.batch_free(b[4])


def test_groups_beyond_the_kernel_arguments_carry_their_arena_records_in_device_memory(ctx):
    """Up to 128 arenas ride in a dispatch's kernel arguments; a larger group (many small arenas: the candidate files of a wide
    query, a file's shards on one of 8 GPUs) uploads its records in front of the dispatch and is still ONE dispatch.  300 arenas
    through groups of 1 024 (the default), 129 and 200, few-term and many-term batches, dense and row outputs: the same bits as
    the oracle's."""
    rng = np.random.default_rng(15)
    plans, vocab = [], None
    for n_blocks in (70, 3, 129, 1, 65):
        plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.02)
        plans.append((plan, ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)))
    arenas = [ctx.arena_load(w, p.desc) for p, w in plans]
    nbs = [p.n_blocks for p, _ in plans]
    order = [int(i) for i in rng.integers(0, len(arenas), size=300)]
    try:
        for n_queries, words in ((300, vocab), (40, vocab[:10])):
            cb = Q.compile_queries([None] + [H.random_expression(rng, words, None) for _ in range(n_queries)])
            ops, poff, _ = cb.arrays()
            terms = H.gpu_terms(ctx, cb)
            bid = ctx.batch_create(terms, ops, poff)
            wants = [O.probe_batch(w, p.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff) for p, w in plans]
            for limit in (0, 129, 200):
                ctx.set_probe_group(limit)
                ctx.timing_read(reset=True)
                got = ctx.probe_many([arenas[i] for i in order], bid, _lib.PROBE_TIMED, cb.n_queries, [nbs[i] for i in order])
                for g, i in zip(got, order):
                    assert np.array_equal(g, wants[i]), (len(terms), limit, i)
                t = ctx.timing_read()
                assert t.n_probes == {0: 1, 129: 3, 200: 2}[limit] and t.n_probe_arenas == 300      # one dispatch per group, whatever its size
            # and as survivor rows
            from bloomsearch_amd.gpu import rows_to_dense
            NQ = cb.n_queries
            Gs = [(nbs[i] + 63) // 64 for i in order]
            rows = ctx.pinned_array(NQ * sum(Gs) * 8).view(np.uint64)
            hdr = ctx.pinned_array(NQ * len(order) * 4).view(np.uint32)
            ctx.set_probe_group(0)
            ctx.probe_many_rows([arenas[i] for i in order], bid, rows, hdr)
            o = 0
            for j, i in enumerate(order):
                assert np.array_equal(rows_to_dense(hdr[j * NQ: (j + 1) * NQ], rows[o: o + NQ * Gs[j]], nbs[i]), wants[i]), (j, i)
                o += NQ * Gs[j]
            ctx.pinned_free(rows.view(np.uint8))
            ctx.pinned_free(hdr.view(np.uint8))
            ctx.batch_free(bid)
    finally:
        ctx.set_probe_group(0)
    for a in arenas:
        ctx.arena_free(a)


def test_survivors_to_device_pointer_and_async_host_output(ctx):
    rng = np.random.default_rng(6)
    plan, _, vocab = H.make_random_arena(rng, 150, absent_frac=0.0)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    cb = Q.compile_queries([H.random_expression(rng, vocab, None) for _ in range(300)])
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    bid = ctx.batch_create(terms, ops, poff)
    aid = ctx.arena_load(words, plan.desc)
    want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    n = 40                                                     # two groups
    G = want.shape[1]
    dmem = HipMem(n * cb.n_queries * G * 8)
    ctx.probe_many_dev([aid] * n, bid, dmem.ptr.value)
    got = dmem.read().reshape(n, cb.n_queries, G)
    dmem.free()
    assert all(np.array_equal(got[i], want) for i in range(n))
    pinned = ctx.pinned_array(n * cb.n_queries * G * 8).view(np.uint64)
    pinned[:] = 0
    ctx.probe_many_into([aid] * n, bid, pinned, _lib.PROBE_ASYNC)
    ctx.sync()
    got = pinned.reshape(n, cb.n_queries, G)
    assert all(np.array_equal(got[i], want) for i in range(n))
    # the latency path (one group, small synchronous result, optional spin wait) returns the same bits
    small = np.zeros(cb.n_queries * G, dtype=np.uint64)
    for spin in (0, 50):
        ctx.set_spin_wait(spin)
        small[:] = 0
        ctx.probe_many_into([aid], bid, small)
        assert np.array_equal(small.reshape(cb.n_queries, G), want)
    ctx.set_spin_wait(0)
    ctx.pinned_free(pinned.view(np.uint8))
    ctx.arena_free(aid)
    ctx.batch_free(bid)


def test_gather_regime_equals_streamed_probe(ctx):
    """Q = 1: few probes against large bitsets read <= terms x k sectors instead of streaming the bitset into LDS
    (bsg_set_gather_cost); the verdicts cannot depend on which way the bits were fetched."""
    blocks = gen_blocks(range(40), rows=4000)
    plan = plan_blocks(blocks, 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aid = ctx.arena_load(words, plan.desc)
    d = synth.draws(0, 4000)
    for exprs in ([Q.And(Q.FieldToken("level", "error"), Q.FieldToken("user_id", str(int(d["user_id"][17]))))],
                  [Q.FieldToken("user_id", str(int(u))) for u in d["user_id"][:5]] + [Q.Token("absent"), Q.Field("nested.az")]):
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
        try:
            ctx.set_lab(3, 0)                                  # the two-kernel path, not the one-dispatch one
            for cost in (0, 256, 1 << 20):
                ctx.set_gather_cost(cost)
                assert np.array_equal(ctx.probe(aid, 40, terms, ops, poff), want), cost
            ctx.set_lab(3, 16)
            assert np.array_equal(ctx.probe(aid, 40, terms, ops, poff), want)
        finally:
            ctx.set_gather_cost(256)
            ctx.set_lab(3, 16)
    ctx.arena_free(aid)


def test_one_dispatch_path_for_small_batches(ctx):
    """k_probe_direct: a synchronous batch of <= 256 queries with a few distinct terms is tested AND evaluated by one
    dispatch, survivors written into page-locked host memory (or a device pointer).  Same bits as the oracle and as the
    two-kernel path: 1 / 7 / 256 queries, groups of 1 and of 9 arenas of different sizes, nil filters, an arena without
    blocks, empty / nil / unknown expressions."""
    rng = np.random.default_rng(77)
    plans, vocab = [], None
    for n_blocks in (1, 64, 65, 130, 200, 3, 70, 129, 31):
        plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.1, max_tokens=400, vocab_size=60)
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        plans.append((plan, words, ctx.arena_load(words, plan.desc)))
    empty = ctx.arena_load(np.zeros(2, dtype=np.uint64), np.zeros(0, dtype=DESC_DTYPE))
    t_before = ctx.timing_read(reset=True)
    for n_queries in (1, 7, 256):
        leaves = [Q.Field("f1"), Q.Field("f44"), Q.Token(vocab[0]), Q.Token(vocab[1]), Q.Token("absent"), Q.FieldToken("f2", vocab[2]),
                  Q.FieldToken("f3", vocab[0]), Q.FieldToken("f1", "absent"), {"ExpressionType": "CONDITION", "Condition": None},
                  {"ExpressionType": "XOR", "Children": []}]

        def small_expr(depth=0):
            if depth >= 3 or rng.random() < 0.4:
                return leaves[int(rng.integers(0, len(leaves)))]
            kids = [small_expr(depth + 1) for _ in range(int(rng.integers(0, 5)))]
            return Q.And(*kids) if rng.random() < 0.5 else Q.Or(*kids)
        exprs = [small_expr() for _ in range(n_queries)]
        if n_queries == 7:
            exprs[0] = None
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        assert len(terms) <= 16
        try:
            bid = ctx.batch_create(terms, ops, poff)
            wants = [O.probe_batch(wd, pl.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff) for pl, wd, _ in plans]
            for ids in ([2], [0, 1, 2, 3, 4, 5, 6, 7, 8], [8, 8, 3]):
                arenas = [plans[i][2] for i in ids]
                nbs = [plans[i][0].n_blocks for i in ids]
                if len(ids) == 3:
                    arenas.insert(1, empty); nbs.insert(1, 0)
                direct = ctx.probe_many(arenas, bid, _lib.PROBE_TIMED, cb.n_queries, nbs)
                plain = ctx.probe_many(arenas, bid, _lib.PROBE_NOFUSE, cb.n_queries, nbs)
                k = 0
                for a, (g, p2) in enumerate(zip(direct, plain)):
                    if nbs[a] == 0:
                        assert g.shape[1] == 0
                        continue
                    assert np.array_equal(g, wants[ids[k]]), (n_queries, ids, a)
                    assert np.array_equal(p2, wants[ids[k]]), (n_queries, ids, a)
                    k += 1
            # survivors left at a device pointer
            G = wants[4].shape[1]
            dmem = HipMem(cb.n_queries * G * 8)
            ctx.probe_many_dev([plans[4][2]], bid, dmem.ptr.value)
            assert np.array_equal(dmem.read().reshape(cb.n_queries, G), wants[4])
            dmem.free()
            ctx.batch_free(bid)
        finally:
            ctx.set_lab(3, 16)
    t = ctx.timing_read()
    assert t.n_fused >= 9 and t.n_probes == 0        # every timed call above was ONE dispatch
    # a batch without a single term (nil and unknown expressions only)
    cb = Q.compile_queries([None, {"ExpressionType": "XOR", "Children": []}, Q.And(), Q.Or()])
    ops, poff, _ = cb.arrays()
    terms = H.gpu_terms(ctx, cb)
    assert len(terms) == 0
    pl, wd, aid0 = plans[3]
    want = O.probe_batch(wd, pl.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
    assert np.array_equal(ctx.probe(aid0, pl.n_blocks, terms, ops, poff), want)
    # k = 17 (fpr 1e-5): more locations than one trip of word reads holds; k = 1 (fpr 0.6)
    for fpr in (1e-5, 0.6):
        plan, _, vocab = H.make_random_arena(rng, 90, fpr=fpr, absent_frac=0.05, max_tokens=300, vocab_size=40)
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        assert int(plan.desc["k"].max()) == 17 if fpr < 0.1 else int(plan.desc["k"].max()) <= 2
        cb = Q.compile_queries([Q.And(Q.Token(vocab[0]), Q.Or(Q.Token(vocab[1]), Q.FieldToken("f1", vocab[2]))), Q.Token(vocab[3]), Q.Field("f7")])
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
        aid = ctx.arena_load(words, plan.desc)
        assert np.array_equal(ctx.probe(aid, 90, terms, ops, poff), want)
        ctx.arena_free(aid)
    ctx.arena_free(empty)
    for _, _, a in plans:
        ctx.arena_free(a)


def test_error_scopes_keep_their_own_message(ctx):
    """bsg_scope_open: every scope owns its error slot, readable from any thread (a goroutine may change OS thread
    between the failing cgo call and bsg_last_error)."""
    import threading
    a, b = ctx.scope(), ctx.scope()
    with pytest.raises(BloomGpuError) as ea:
        a.arena_free(0xDEAD)
    with pytest.raises(BloomGpuError) as eb:
        b.batch_free(0xBEEF)
    assert "arena" in str(ea.value) and "batch" in str(eb.value)
    seen = {}

    def reader():                      # another OS thread reads both slots
        seen["a"] = a.L.bsg_last_error(a.h).decode()
        seen["b"] = b.L.bsg_last_error(b.h).decode()
    th = threading.Thread(target=reader)
    th.start(); th.join()
    assert "arena" in seen["a"] and "57005" in seen["a"] and "batch" in seen["b"]
    # scopes are full aliases of the context: compute calls work through them
    assert a.hash_strings(["hello"]).shape == (1, 4)
    a.close(); b.close()


def _sections_of(ctx, rng, n_blocks, max_tokens=2500):
    plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.05, max_tokens=max_tokens)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    sections = []
    for b in range(n_blocks):
        fl = []
        for c in range(3):
            d = plan.desc[b * 3 + c]
            nw = O.words_for(int(d["m"])) if d["m"] else 0
            fl.append(O.Filter(int(d["m"]), int(d["k"]), words[int(d["word_off"]): int(d["word_off"]) + nw]) if d["m"] else None)
        sections.append(O.encode_filter_section(fl))
    return plan, words, vocab, sections


def test_region_cursor_streams_chunks_like_blockFilterCursor():
    """a9 (file_format.go:511-662): a file's filter region handed over in chunks — in order with sections straddling the
    chunk boundaries, out of order, with a gap that is never read, with a corrupt section — on a single-device context
    and on one that shards over three entries.  Decoded arenas must probe exactly like the oracle over the same filters;
    unread sections report -7 and behave as nil filters; a corrupt one is isolated."""
    rng = np.random.default_rng(123)
    n_blocks = 61
    with Context((0,)) as c1, Context(device_ids(3)) as c3:
        plan, words, vocab, sections = _sections_of(c1, rng, n_blocks)
        cb = Q.compile_queries([None] + [H.random_expression(rng, vocab, None) for _ in range(300)])
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(c1, cb)
        # a "file": 1 000 bytes of row data, then the sections back to back, except a 70-byte hole before block 30
        file_bytes = bytearray(rng.integers(0, 256, size=1000, dtype=np.uint8).tobytes())
        begin, end = [], []
        for b, s in enumerate(sections):
            if b == 30:
                file_bytes += b"\xAA" * 70
            begin.append(len(file_bytes))
            file_bytes += s
            end.append(len(file_bytes))
        file_bytes += b"footer"
        file_bytes = bytes(file_bytes)
        want_all = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)

        def expect(nil_blocks):
            d2 = plan.desc.copy()
            for b in nil_blocks:
                d2["m"][b * 3: b * 3 + 3] = 0
            return O.probe_batch(words, d2.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)

        for ctx in (c1, c3):
            # (1) in order, chunk sizes that never line up with section boundaries (incl. chunks smaller than a header)
            sid = ctx.arena_stream_begin(begin, end)
            o = 900
            sizes = [7, 1, 33, 5000, 11, 40000, 3, 100000]
            i = 0
            while o < len(file_bytes):
                n = sizes[i % len(sizes)]; i += 1
                ctx.arena_stream_append(sid, o, file_bytes[o: o + n])
                o += n
            aid, status = ctx.arena_stream_finish(sid, n_blocks)
            assert not status.any()
            assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), want_all)
            ctx.arena_free(aid)
            # (2) out of order + a range that is never read (blocks 10..14) + a chunk appended twice
            sid = ctx.arena_stream_begin(begin, end)
            ranges = [(begin[40], len(file_bytes)), (begin[15], begin[40] + 9), (0, begin[10] + 5), (begin[15], begin[20])]
            for lo, hi in ranges:
                ctx.arena_stream_append(sid, lo, file_bytes[lo:hi])
            aid, status = ctx.arena_stream_finish(sid, n_blocks)
            unread = {b for b in range(10, 15) if len(sections[b]) > 0}
            assert {b for b in range(n_blocks) if status[b] == -7} == unread
            assert not any(status[b] for b in range(n_blocks) if b not in unread)
            assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), expect(unread))
            ctx.arena_free(aid)
            # (3) a flipped bit inside block 22's section: ErrInvalidHash for that block only
            bad = bytearray(file_bytes); bad[(begin[22] + end[22]) // 2] ^= 4; bad = bytes(bad)
            sid = ctx.arena_stream_begin(begin, end)
            for o in range(0, len(bad), 65536):
                ctx.arena_stream_append(sid, o, bad[o: o + 65536])
            aid, status = ctx.arena_stream_finish(sid, n_blocks)
            assert status[22] == -2 and not any(status[b] for b in range(n_blocks) if b != 22)
            assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), expect({22}))
            ctx.arena_free(aid)
            # (4) abort releases everything; the whole-region form on the sharded context equals the single-device one
            sid = ctx.arena_stream_begin(begin, end)
            ctx.arena_stream_append(sid, begin[0], file_bytes[begin[0]: begin[5]])
            ctx.arena_stream_abort(sid)
            with pytest.raises(BloomGpuError):
                ctx.arena_stream_finish(sid, n_blocks)
            aid, status = ctx.arena_load_sections(sections)
            assert not status.any()
            assert np.array_equal(ctx.probe(aid, n_blocks, terms, ops, poff), want_all)
            ctx.arena_free(aid)


@pytest.mark.parametrize("n_terms", [129, 191, 192, 193, 700, 4100])
def test_many_term_mode_every_compaction_depth(ctx, n_terms):
    """The many-term probe mode (> 128 distinct terms of one kind: per-wave rounds with ballot compaction, padded tail of
    the last term word probed like real terms) against the oracle at term counts around the 64-term word boundaries, for
    filters with k = 1, 2, 7, 10, 14 and for 0..3 compaction rounds, staged and gathered."""
    rng = np.random.default_rng(n_terms)
    vocab = ["tok%d" % i for i in range(6000)]
    blocks = []
    for b in range(13):
        toks = sorted({vocab[i] for i in rng.integers(0, len(vocab), size=int(rng.integers(1, 2500)))})
        blocks.append(H.entry_sets_from_strings(["f"], toks, ["f::" + t for t in toks]))
    for fpr in (0.5, 0.3, 0.01, 0.001, 0.0001):          # k = 1, 2, 7, 10, 14
        plan = plan_blocks(blocks, fpr)
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        picks = rng.choice(len(vocab), size=n_terms, replace=False)
        exprs = [Q.Token(vocab[i]) for i in picks] + [Q.And(Q.Token(vocab[picks[0]]), Q.FieldToken("f", vocab[picks[1]]))]
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
        aid = ctx.arena_load(words, plan.desc)
        try:
            for rounds in (0, 1, 2, 3):
                ctx.set_lab(1, rounds)
                for cost in (256, 1 << 24):               # staged, then everything gathered from global memory
                    ctx.set_gather_cost(cost)
                    assert np.array_equal(ctx.probe(aid, 13, terms, ops, poff), want), (fpr, rounds, cost)
        finally:
            ctx.set_lab(1, 2)
            ctx.set_gather_cost(256)
            ctx.arena_free(aid)


def test_shards_that_disagree_on_the_one_dispatch_path():
    """A context over two devices where device 0 holds 65 shards (two launch groups: streaming kernels + an asynchronous
    copy) and device 1 holds one (k_probe_direct + doorbell): the doorbell of device 1 must not excuse device 0 from its
    stream wait (round-2 advice: survivors were read before their copy had landed)."""
    rng = np.random.default_rng(5)
    with Context(device_ids(2)) as mctx:
        plans = []
        for i in range(65):
            n_blocks = 2 if i == 40 else 1                     # only arena 40 has a block for device 1
            plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.0, max_tokens=300, vocab_size=50)
            words = mctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
            plans.append((plan, words, mctx.arena_load(words, plan.desc)))
        exprs = [Q.Token(vocab[i]) for i in range(6)] + [Q.And(Q.Token(vocab[0]), Q.Field("f1")), None]
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(mctx, cb)
        assert len(terms) <= 16
        bid = mctx.batch_create(terms, ops, poff)
        for _ in range(20):                                     # the race needed the copy to lose against the bell
            got = mctx.probe_many([p[2] for p in plans], bid, 0, cb.n_queries, [p[0].n_blocks for p in plans])
            for (pl, wd, _), g in zip(plans, got):
                assert np.array_equal(g, O.survivors_tree(wd, pl.desc.view(O.DESC_DTYPE), exprs))
        mctx.batch_free(bid)


def test_bsg_query_one_call_strings_in_survivors_out(ctx):
    """bsg_query: the term STRINGS go in, hashed on the host by the kernels' own base_hashes; for a handful of terms the hashes
    and programs ride in the kernel arguments of one dispatch (k_query_direct) — no batch object, nothing uploaded.  Same bits as
    the tree-walking oracle and as hash + batch_create + probe_many, for the one-dispatch shape and for everything that falls
    back to the batch path inside the call (17+ terms, long programs, 33+ arenas), on single- and multi-device contexts."""
    rng = np.random.default_rng(2026)
    plans, vocab = [], None
    for n_blocks in (1, 64, 65, 130, 200, 3):
        plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.1, max_tokens=400, vocab_size=60)
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        plans.append((plan, words))
    unicode_tok = [t for t in vocab if not t.isascii()][0]
    shapes = {
        "one 3-term query": [Q.And(Q.FieldToken("f1", vocab[0]), Q.Token(vocab[1]), Q.Field("f2"))],
        "nil query": [None],
        "no terms at all": [Q.And(), Q.Or(), {"ExpressionType": "XOR"}],
        "seven queries": [Q.Token(vocab[i]) for i in range(5)] + [Q.Or(Q.Token("absent"), Q.Field("f3"), Q.Token(unicode_tok)), None],
        "16 distinct terms": [Q.Or(*[Q.Token(vocab[i]) for i in range(16)])],
        "17 distinct terms (batch path)": [Q.Or(*[Q.Token(vocab[i]) for i in range(17)])],
        "long programs (batch path)": [Q.And(*[Q.Or(Q.Token(vocab[i % 8]), Q.Field("f%d" % (i % 5))) for i in range(40)]) for _ in range(4)],
        "many queries (batch path)": [Q.Token(vocab[i % 12]) for i in range(300)],
    }
    for n_dev in (1, 3):
        with Context(device_ids(n_dev)) as c:
            ids = [c.arena_load(w, p.desc) for p, w in plans]
            empty = c.arena_load(np.zeros(2, dtype=np.uint64), np.zeros(0, dtype=DESC_DTYPE))
            nbs = [p.n_blocks for p, _ in plans]
            for name, exprs in shapes.items():
                cb = Q.compile_queries(exprs)
                got = c.query(ids[:2] + [empty] + ids[2:], nbs[:2] + [0] + nbs[2:], cb)
                del got[2]
                for (p, w), g in zip(plans, got):
                    assert np.array_equal(g, O.survivors_tree(w, p.desc.view(O.DESC_DTYPE), exprs)), (n_dev, name)
            # 33 arenas on one device: beyond what one k_query_direct covers -> the batch path, same bits
            many_ids = (ids * 6)[:33] if n_dev == 1 else (ids * 20)[:100]
            many_nbs = (nbs * 20)[: len(many_ids)]
            cb = Q.compile_queries(shapes["one 3-term query"])
            got = c.query(many_ids, many_nbs, cb)
            for i, g in enumerate(got):
                p, w = plans[i % len(plans)]
                assert np.array_equal(g, O.survivors_tree(w, p.desc.view(O.DESC_DTYPE), shapes["one 3-term query"]))
            # malformed input is rejected before any launch
            with pytest.raises(BloomGpuError):
                bad = Q.compile_queries([Q.Token("x")])
                bad._packed = (np.zeros(1, np.uint8), np.asarray([0, 1], np.uint32), np.asarray([7], np.uint32), *bad.arrays()[:2])
                c.query(ids[:1], nbs[:1], bad)


def test_batches_beyond_one_launch_are_split_inside_the_library(ctx):
    """The reference's evaluator has no batch limit (query_exec.go:89-126).  One probe launch holds ~22 000 distinct terms of a
    kind and ~100 verdict words per 256-query chunk: a batch beyond either is cut into runs of queries inside bsg_batch_create
    and probed run by run — the caller sees one batch and one result, bit-identical to the tree-walking oracle."""
    rng = np.random.default_rng(31)
    plan, blocks_str, vocab = H.make_random_arena(rng, 70, absent_frac=0.05, max_tokens=500, vocab_size=40000)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aid = ctx.arena_load(words, plan.desc)
    aid2 = ctx.arena_load(words, plan.desc)
    members = [t for b in range(0, 70, 7) for t in blocks_str[b][1][:20]]
    cases = {
        # 30 000 distinct Token terms (> 22 000 of one kind), one or two per query
        "30000 terms": [Q.Token(vocab[i]) for i in range(30000)] + [Q.And(Q.Token(m), Q.Token(vocab[7])) for m in members],
        # 256 queries x 40 terms spread over 10 240 distinct terms: 160 verdict words in one chunk
        "160 verdict words per chunk": [Q.Or(*[Q.Token(vocab[(q * 40 + j) % 40000]) for j in range(40)]) for q in range(256)] + [Q.Token(members[0]), None],
    }
    for name, exprs in cases.items():
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        bid = ctx.batch_create(terms, ops, poff)
        want = O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), exprs)
        got = ctx.probe_many([aid, aid2], bid, 0, cb.n_queries, [70, 70])
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want), name
        assert np.array_equal(ctx.probe_batch(aid, bid, cb.n_queries, 70), want), name
        ctx.batch_free(bid)
        assert want.any()
        # and through the one-call entry point (falls back to the same composite inside)
        assert np.array_equal(ctx.query([aid], [70], cb)[0], want), name
    # a SINGLE query beyond the limits stays unsupported, with a message
    cb = Q.compile_queries([Q.Or(*[Q.Token(vocab[i]) for i in range(23000)])])
    ops, poff, _ = cb.arrays()
    with pytest.raises(BloomGpuError) as e:
        ctx.batch_create(H.gpu_terms(ctx, cb), ops, poff)
    assert e.value.code == _lib.BSG_E_UNSUPPORTED
    # survivor lists: ascending block indices == the set bits
    from bloomsearch_amd.gpu import survivor_list
    row = want[0]
    assert survivor_list(row, 70).tolist() == [b for b in range(70) if (int(row[b >> 6]) >> (b & 63)) & 1]
    ctx.arena_free(aid)
    ctx.arena_free(aid2)


def test_a_batch_without_terms_against_a_group_beyond_the_kernel_arguments(ctx):
    """Queries that reference no term (nil queries, nil conditions: always true — query_exec.go:81-83, 101-104) launch no probe
    kernel; their evaluation must still find the records of a group of more than 128 arenas (device-memory table), whatever a
    previous, larger call left in that table."""
    rng = np.random.default_rng(25)
    plans = []
    for n_blocks in (200, 3, 65, 130):
        plan, _, vocab = H.make_random_arena(rng, n_blocks, absent_frac=0.0)
        plans.append((plan, ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)))
    arenas = [ctx.arena_load(w, p.desc) for p, w in plans]
    nbs = [p.n_blocks for p, _ in plans]
    # a larger call first: its records stay in the slot's table
    cb0 = Q.compile_queries([H.random_expression(rng, vocab, None) for _ in range(20)])
    ops0, poff0, _ = cb0.arrays()
    bid0 = ctx.batch_create(H.gpu_terms(ctx, cb0), ops0, poff0)
    ctx.probe_many([arenas[0]] * 400, bid0, 0, cb0.n_queries, [nbs[0]] * 400)
    ctx.batch_free(bid0)
    cb = Q.compile_queries([None, None, {"ExpressionType": "CONDITION", "Condition": None}])
    ops, poff, _ = cb.arrays()
    bid = ctx.batch_create(H.gpu_terms(ctx, cb), ops, poff)
    order = [int(i) for i in rng.integers(1, len(arenas), size=150)]
    got = ctx.probe_many([arenas[i] for i in order], bid, 0, cb.n_queries, [nbs[i] for i in order])
    for g, i in zip(got, order):
        want = np.full((3, (nbs[i] + 63) // 64), ~np.uint64(0), dtype=np.uint64)
        if nbs[i] & 63:
            want[:, -1] = np.uint64((1 << (nbs[i] & 63)) - 1)
        assert np.array_equal(g, want), i
    ctx.batch_free(bid)
    for a in arenas:
        ctx.arena_free(a)
