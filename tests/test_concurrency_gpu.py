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
    with Context((0, 0, 0)) as m:
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
