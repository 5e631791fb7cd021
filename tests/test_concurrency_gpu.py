"""The C-ABI is re-entrant on one context (file-worker goroutines call bsg_probe concurrently — up to
MaxQueryConcurrency of them — while the flush worker builds): many host threads drive one bsg_ctx at once through
every family of entry points, and every result must still equal the oracle's, bit for bit.  ctypes releases the GIL
for the duration of each C call, so the calls really overlap."""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from bloomsearch_amd import ingest as I, query as Q, synth
from oracle import oracle as O
from oracle import walker_oracle as W
from tests import helpers as H
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu

N_THREADS = 12
ROUNDS = int(os.environ.get("BSG_TEST_ROUNDS", "6"))     # raise for a soak run


def _probe_job(ctx, seed):
    rng = np.random.default_rng(seed)
    plan, _, vocab = H.make_random_arena(rng, int(rng.integers(3, 150)), absent_frac=0.03)
    cb = Q.compile_queries([None] + [H.random_expression(rng, vocab, None) for _ in range(int(rng.integers(1, 400)))])
    ops, poff, _ = cb.arrays()
    want_words = H.oracle_words(plan)
    want = O.probe_batch(want_words, plan.desc.view(O.DESC_DTYPE), H.oracle_terms(cb).view(O.TERM_DTYPE), ops, poff)
    for _ in range(ROUNDS):
        words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        assert np.array_equal(words, want_words)
        terms = H.gpu_terms(ctx, cb)
        aid = ctx.arena_load(words, plan.desc)
        bid = ctx.batch_create(terms, ops, poff)
        got = ctx.probe_batch(aid, bid, cb.n_queries, plan.n_blocks)
        many = ctx.probe_many([aid, aid, aid], bid, 0, cb.n_queries, [plan.n_blocks] * 3)
        one = ctx.probe(aid, plan.n_blocks, terms, ops, poff)
        ctx.batch_free(bid)
        ctx.arena_free(aid)
        assert np.array_equal(got, want) and np.array_equal(one, want)
        assert all(np.array_equal(m, want) for m in many)
    return True


def _single_query_job(ctx, seed):
    """interactive queries: one synchronous 1- or 2-query batch after the other (k_probe_direct: page-locked result
    buffers from a per-device pool, a completion doorbell) while other threads do the same and more"""
    rng = np.random.default_rng(seed)
    plan, _, vocab = H.make_random_arena(rng, int(rng.integers(1, 300)), absent_frac=0.03, max_tokens=200, vocab_size=30)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aid = ctx.arena_load(words, plan.desc)
    for _ in range(ROUNDS * 40):
        exprs = [Q.And(Q.Token(vocab[int(rng.integers(0, 30))]), Q.Or(Q.FieldToken("f%d" % rng.integers(0, 9), vocab[int(rng.integers(0, 30))]),
                                                                     Q.Field("f%d" % rng.integers(0, 45))))]
        if rng.random() < 0.3:
            exprs.append(Q.Token("absent"))
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.oracle_terms(cb)
        want = O.probe_batch(words, plan.desc.view(O.DESC_DTYPE), terms.view(O.TERM_DTYPE), ops, poff)
        assert np.array_equal(ctx.probe(aid, plan.n_blocks, terms, ops, poff), want)
        assert np.array_equal(ctx.query([aid], [plan.n_blocks], cb)[0], want)        # the one-call path: hashes + programs in the kernel arguments
    ctx.arena_free(aid)
    return True


def _ingest_job(ctx, seed):
    rows = [synth.rows_json(seed * 1000 + b * 300, 300) for b in range(3)]
    sets = []
    for rs in rows:
        s = (set(), set(), set())
        for r in rs:
            W.index_row(r, s)
        sets.append(s)
    for _ in range(ROUNDS):
        res = I.device_ingest(ctx, rows, 0.001, parent_of_set=[0, 0, 0], n_parents=1, flags=seed & 1)
        for i, s in enumerate(sets):
            for kind in range(3):
                assert int(res.counts[i, kind]) == len(s[kind])
                want = O.build_sized(sorted(s[kind]), 0.001)
                assert np.array_equal(res.filter_words(i, kind), want.words)
        union = [set().union(*(s[k] for s in sets)) for k in range(3)]
        assert [int(x) for x in res.counts[3]] == [len(u) for u in union]
    return True


def _match_job(ctx, seed):
    rows = synth.rows_json(seed * 777, 2000)
    d = synth.draws(seed * 777, 2000)
    e = Q.Or(Q.And(Q.FieldToken("level", "warn"), Q.FieldToken("nested.az", "az-1")), Q.Token("region-7"))
    want = ((d["level"] == synth.LEVELS.index("warn")) & (d["az"] == 1)) | (d["region"] == 7)
    for _ in range(ROUNDS):
        got, handed_back = ctx.match_rows(rows, Q.CompiledMatcher(e))
        assert len(handed_back) == 0 and np.array_equal(got, want)
    return True


def test_many_threads_one_context(ctx):
    errors = []
    barrier = threading.Barrier(N_THREADS)

    def run(i):
        try:
            barrier.wait(timeout=60)
            return (_probe_job, _single_query_job, _match_job, _ingest_job, _single_query_job, _probe_job)[i % 6](ctx, 100 + i)
        except BaseException as exc:  # noqa: BLE001 - reported below with the thread index
            errors.append((i, repr(exc)))
            return False

    with ThreadPoolExecutor(N_THREADS) as pool:
        results = list(pool.map(run, range(N_THREADS)))
    assert not errors, errors
    assert all(results)


def test_many_threads_one_sharded_context():
    """The same mix on a context of three entries whose construct / match calls are all cut into parts (threads inside the
    library on top of the callers' threads): per-device locks are only ever taken one at a time, so nothing can deadlock, and
    every result still equals the oracle's."""
    from bloomsearch_amd.gpu import Context
    errors = []
    with Context(device_ids(3)) as m:
        m.set_lab(7, 1)
        m.set_lab(8, 1)
        barrier = threading.Barrier(9)

        def run(i):
            try:
                barrier.wait(timeout=60)
                return (_probe_job, _ingest_job, _match_job, _single_query_job, _ingest_job, _probe_job, _match_job, _ingest_job, _single_query_job)[i % 9](m, 300 + i)
            except BaseException as exc:  # noqa: BLE001
                errors.append((i, repr(exc)))
                return False

        with ThreadPoolExecutor(9) as pool:
            results = list(pool.map(run, range(9)))
        assert not errors, errors
        assert all(results)
        assert (m.device_calls() > 0).all()


# ---- concurrent bsg_query callers share dispatches (csrc/combine_api.inc) ----

def _combiner_case(ctx, seed, n_threads, rounds):
    """Python threads, each issuing bsg_query calls of random queries over random arena subsets — the same few arenas and a small
    pool of queries, so calls meet on arena lists and on query sets, while others differ in both — every result against the oracle's
    tree-walking evaluator."""
    rng = np.random.default_rng(seed)
    plans, words, aids = [], [], []
    vocab = None
    for nb in (1, 63, 64, 130, 257, 1000):
        plan, _, vocab = H.make_random_arena(rng, nb, absent_frac=0.03, max_tokens=120, vocab_size=40)
        w = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        plans.append(plan); words.append(w); aids.append(ctx.arena_load(w, plan.desc))
    pool = [Q.And(Q.Token(vocab[i % 40]), Q.Or(Q.FieldToken("f%d" % (i % 9), vocab[(i * 7) % 40]), Q.Field("f%d" % (i % 45)))) for i in range(12)]
    pool += [H.random_expression(rng, vocab[:12], None) for _ in range(12)] + [None]
    want = [[O.survivors_tree(w, p.desc.view(O.DESC_DTYPE), [e])[0] for e in pool] for w, p in zip(words, plans)]
    lists = [[0], [5], [2, 3], [5, 4], [1, 0, 5], [3], [2, 2, 1]]             # (an arena may be named twice by one call)
    errors = []
    gate = threading.Barrier(n_threads)

    def run(t):
        r = np.random.default_rng(seed * 1000 + t)
        try:
            gate.wait(timeout=60)
            for _ in range(rounds):
                sub = lists[int(r.integers(0, len(lists)))]
                qs = [int(x) for x in r.integers(0, len(pool), size=int(r.choice([1, 1, 1, 2, 5])))]
                got = ctx.query([aids[i] for i in sub], [plans[i].n_blocks for i in sub], Q.compile_queries([pool[q] for q in qs]))
                for g, i in zip(got, sub):
                    for row, q in zip(g, qs):
                        assert np.array_equal(row, want[i][q]), "thread %d: arena %d query %d differs from the oracle" % (t, i, q)
        except BaseException as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    ths = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for a in aids:
        ctx.arena_free(a)
    assert not errors, errors[:3]


def test_concurrent_queries_equal_the_oracle_whatever_they_are_merged_with(ctx):
    # Python threads rarely meet inside the library by themselves (the interpreter lock spaces their calls out): the collector is told
    # to wait up to 2 ms for 8 queued calls (lab key 15), so every cycle merges a random handful of arena lists and query sets
    ctx.set_lab(15, (2000 << 16) | 8)
    try:
        ctx.query_stats(reset=True)
        _combiner_case(ctx, 11, 24, ROUNDS * 10)
        st = ctx.query_stats()
    finally:
        ctx.set_lab(15, 0)
    # (a random expression beyond 16 terms / 128 program words goes alone without ever entering the combiner)
    assert 0.9 * 24 * ROUNDS * 10 <= st["calls"] <= 24 * ROUNDS * 10 and st["cycle_calls"] == st["calls"]
    assert st["max_calls_per_cycle"] >= 6 and st["dispatches"] > 0, "calls did not share cycles: %r" % (st,)


def test_short_job_lists_in_the_kernel_arguments_and_uploaded(ctx):
    # cycles of <= 4 calls: their job table fits the kernel arguments (k_query_jobs_inline); with key 21 = 0 the same lists are uploaded
    for inline in (1, 0):
        ctx.set_lab(21, inline)
        ctx.set_lab(15, (1000 << 16) | 4)
        try:
            _combiner_case(ctx, 13 + inline, 8, ROUNDS * 4)
        finally:
            ctx.set_lab(15, 0)
            ctx.set_lab(21, 1)


def test_a_cycle_beyond_its_scratch_budget_is_served_in_parts(ctx):
    # key 24: a part holds at most 512 bytes of survivor rows, so cycles of 8 calls split into several parts (one call alone in a part
    # when it exceeds the budget by itself)
    ctx.set_lab(24, 512)
    ctx.set_lab(15, (2000 << 16) | 8)
    try:
        _combiner_case(ctx, 21, 16, ROUNDS * 4)
    finally:
        ctx.set_lab(15, 0)
        ctx.set_lab(24, 64 << 20)


def test_concurrent_queries_on_a_sharded_context():
    from bloomsearch_amd.gpu import Context
    with Context(device_ids(3)) as m:
        m.set_lab(15, (2000 << 16) | 6)
        _combiner_case(m, 12, 12, ROUNDS * 8)
        st = m.query_stats()
        assert st["max_calls_per_cycle"] >= 4 and st["dispatches"] > 0, st


def test_native_callers_share_dispatches_bit_exactly(ctx):
    """T native threads (no interpreter lock between them) x bsg_query of one 3-term query against 1 and 3 arenas: every result
    compared inside the driver with the rows the solo path returned — which are checked against the oracle here."""
    from bloomsearch_amd import conc
    blocks = [synth.block_entry_sets(b * 400, 400) for b in range(200)]
    from bloomsearch_amd.arena import plan_blocks
    plan = plan_blocks(blocks, 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aids = [ctx.arena_load(words, plan.desc) for _ in range(6)]
    exprs = synth.make_queries(48, "c2", seed=77)
    ctx.set_lab(12, 0)
    expected = np.stack([ctx.query([aids[0]], [200], Q.compile_queries([e]))[0][0] for e in exprs])
    assert np.array_equal(expected, O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), exprs))
    ctx.query_stats(reset=True)
    off = conc.run(ctx, exprs, aids, 200, expected, n_threads=32, seconds=0.3)
    st_off = ctx.query_stats()
    assert off["mismatches"] == 0 and off["errors"] == 0 and off["calls"] > 0
    assert st_off["calls"] == 0                                     # combining off: no call ever reaches the combiner
    ctx.set_lab(12, 1)
    for apc in (1, 3):
        on = conc.run(ctx, exprs, aids, 200, expected, n_threads=32, seconds=0.3, arenas_per_call=apc)
        st = ctx.query_stats()
        assert on["mismatches"] == 0 and on["errors"] == 0 and on["calls"] > 0
        assert st["calls"] == on["calls"] + 32 == st["cycle_calls"]          # (+ every thread's untimed first call)
        assert st["max_calls_per_cycle"] > 4 and st["cycles"] < st["calls"], st
    for a in aids:
        ctx.arena_free(a)


def test_a_freed_arena_id_is_forgotten_by_the_callers_caches(ctx):
    """bsg_query looks arena ids up in a per-thread cache (the shared table is a cache line 256 callers would write): freeing an id must
    reach every thread's cache — the next call fails with NOTFOUND instead of probing freed memory — and a new arena is found."""
    from bloomsearch_amd._lib import BloomGpuError
    rng = np.random.default_rng(5)
    plans = [H.make_random_arena(rng, nb, absent_frac=0.03, max_tokens=60, vocab_size=20) for nb in (70, 130)]
    vocab = plans[0][2]
    exprs = [Q.Token(vocab[3]), Q.Or(Q.Token(vocab[5]), Q.Field("f1")), None]
    cb = Q.compile_queries(exprs)
    words = [ctx.build(p.blob, p.off, p.fstart, p.desc, p.n_words) for p, _, _ in plans]
    want = [O.survivors_tree(w, p.desc.view(O.DESC_DTYPE), exprs) for w, (p, _, _) in zip(words, plans)]
    a0 = ctx.arena_load(words[0], plans[0][0].desc)
    seen = []

    def other_thread():                                    # a second thread caches the id too
        seen.append(np.array_equal(np.stack(ctx.query([a0], [70], cb)[0]), want[0]))

    t = threading.Thread(target=other_thread); t.start(); t.join()
    assert seen == [True]
    assert np.array_equal(np.stack(ctx.query([a0], [70], cb)[0]), want[0])
    ctx.arena_free(a0)
    with pytest.raises(BloomGpuError):
        ctx.query([a0], [70], cb)
    a1 = ctx.arena_load(words[1], plans[1][0].desc)
    assert np.array_equal(np.stack(ctx.query([a1], [130], cb)[0]), want[1])
    ctx.arena_free(a1)


def test_combined_queries_beside_a_long_ingest_sleep_instead_of_spinning(ctx):
    """a collector whose dispatch sits behind somebody else's long work on the device (the flush worker's
    bsg_ingest_rows: tens of milliseconds of k_ingest_rows per call at 10 M rows) used to busy-wait 20 ms on its doorbell before it
    fell back to hipStreamSynchronize under the device lock.  Now it polls a few times its usual wait (<= 1 ms; bsg_set_lab key 25), then sleeps on a
    blocking-sync event recorded behind its dispatch.  64 native callers for 1.5 s next to a thread that ingests ~1 GB of rows over
    and over: every result bit-exact, and the whole process (callers + collector + the ingesting thread + the runtime) stays far
    below a processor per caller — the 20 ms spins cost milliseconds of processor time per call."""
    from bloomsearch_amd import conc
    from bloomsearch_amd.arena import plan_blocks
    blocks = [synth.block_entry_sets(b * 400, 400) for b in range(200)]
    plan = plan_blocks(blocks, 0.001)
    words = ctx.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
    aids = [ctx.arena_load(words, plan.desc) for _ in range(6)]
    exprs = synth.make_queries(48, "c2", seed=78)
    ctx.set_lab(12, 0)
    expected = np.stack([ctx.query([aids[0]], [200], Q.compile_queries([e]))[0][0] for e in exprs])
    assert np.array_equal(expected, O.survivors_tree(words, plan.desc.view(O.DESC_DTYPE), exprs))
    ctx.set_lab(12, 1)
    # ~1 GB of JSON rows (4 000 distinct rows, tiled): k_ingest_rows runs ~10 ms per call, back to back
    base = synth.rows_json(0, 4000)
    reps = 1000
    one = np.frombuffer(b"".join(base), dtype=np.uint8)
    blob = np.tile(one, reps)
    lens = np.tile(np.asarray([len(r) for r in base], dtype=np.uint64), reps)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    n_rows = len(lens)
    first = np.asarray([0, n_rows], dtype=np.uint32)
    stop, ingests, failures = threading.Event(), [], []

    def flush_worker():
        try:
            while not stop.is_set():
                ing = ctx.ingest_rows((blob, off), first, np.zeros(1, dtype=np.uint32), 1, flags=1)
                ingests.append(ctx.ingest_stats(ing).ms_walk)
                ctx.ingest_free(ing)
        except Exception as exc:  # noqa: BLE001 - reported by the assertion below
            failures.append(repr(exc))

    quiet = conc.run(ctx, exprs, aids, 200, expected, n_threads=64, seconds=0.5)
    t = threading.Thread(target=flush_worker)
    t.start()
    try:
        while not ingests and not failures and t.is_alive():
            threading.Event().wait(0.05)                 # the first call has uploaded its rows: the device is busy from here on
        busy = conc.run(ctx, exprs, aids, 200, expected, n_threads=64, seconds=1.5)
    finally:
        stop.set()
        t.join()
    assert not failures, failures
    assert len(ingests) >= 2 and max(ingests) > 5.0, ingests          # milliseconds of k_ingest_rows per call: the device WAS busy
    assert quiet["mismatches"] == 0 and quiet["errors"] == 0
    assert busy["mismatches"] == 0 and busy["errors"] == 0 and busy["calls"] > 200
    # processor time of the whole process per wall second: 64 callers that spin would hold ~their quota; sleeping ones a few CPUs
    assert busy["cpus_busy"] < 8.0, busy
    assert busy["p99_us"] < 200_000, busy
    for a in aids:
        ctx.arena_free(a)
