"""The write side and the final row test on a context over several devices ("Build: shards naturally by block
as well"; partitions are independent in flush.go:191-254, surviving blocks in query_exec.go:729-764).

A Go host is ONE process that opens all of a node's GPUs.  bsg_hash_entries, bsg_build*, bsg_build_sections,
bsg_ingest_* and bsg_match_rows cut a call that is large enough into one part per device (contiguous runs of entries /
filters / sets / rows, each on a thread of its own; a parent whose children sit on several devices is unioned per device
first and the partials are merged on one device) and give a small call ONE device, chosen round-robin among those nobody
holds.  This box has one GPU, so the "devices" of these contexts are entries that all name device 0 — every host path
(partitioning, per-part staging, the dense hand-over of partial parents, the assembly of words / sections / resident
arenas) is the one a node with 8 GPUs runs; only the peer copies degenerate into same-device copies.

The bar: hashes, bitsets, section bytes, exact counts, statuses, fallback lists and match bitmaps IDENTICAL to the
single-device context's — and to the oracle's."""
import threading

import numpy as np
import pytest

from bloomsearch_amd import host as Hst, ingest as I, query as Q, synth
from bloomsearch_amd._lib import DESC_DTYPE
from bloomsearch_amd.arena import entry_sets_from_strings, plan_blocks
from bloomsearch_amd.gpu import Context
from oracle import oracle as O
from oracle import walker_oracle as W
from tests import helpers as H
from tests.helpers import device_ids

pytestmark = pytest.mark.gpu

FPR = 0.001


def sharded(n):
    c = Context(device_ids(n))
    c.set_lab(7, 1)        # every construct call is cut over the devices, however small
    c.set_lab(8, 1)
    return c


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_hash_build_and_sections_identical_on_a_sharded_context(ctx, n_dev):
    rng = np.random.default_rng(n_dev)
    blocks = []
    for b in range(37):
        n = int(rng.integers(0, 4000))
        toks = ["t%d_%d" % (b, i) for i in range(n)]
        blocks.append(entry_sets_from_strings(["f%d" % i for i in range(b % 9)], toks, ["f::" + t for t in toks[: n // 2]]))
    blocks.append(entry_sets_from_strings(["only"], ["big%d" % i for i in range(60000)], ["k::v"]))              # 105 KiB: LDS-staged
    blocks.append(entry_sets_from_strings(["only"], ["huge%d" % i for i in range(160000)], ["k::v"]))            # beyond LDS
    blocks.append(entry_sets_from_strings([], [], []))
    plan = plan_blocks(blocks, FPR, absent={(3, 1), (5, 0), (5, 1), (5, 2), (len(blocks) - 1, 2)})
    want = H.oracle_words(plan)
    with sharded(n_dev) as m:
        before = m.device_calls()
        assert np.array_equal(m.hash_entries(plan.blob, plan.off), ctx.hash_entries(plan.blob, plan.off))
        got = m.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        assert np.array_equal(got, want)
        assert np.array_equal(m.build_hashed(ctx.hash_entries(plan.blob, plan.off), plan.fstart, plan.desc, plan.n_words), want)
        try:
            m.set_lab(6, 0)                                      # the bitset beyond LDS binned by window on whichever device got it
            assert np.array_equal(m.build(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words), want)
        finally:
            m.set_lab(6, 4 << 20)
        secs = m.build_sections(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        assert secs == ctx.build_sections(plan.blob, plan.off, plan.fstart, plan.desc, plan.n_words)
        for b in (0, 7, len(blocks) - 3, len(blocks) - 1):
            fl = []
            for c in range(3):
                d = plan.desc[b * 3 + c]
                fl.append(None if int(d["m"]) == 0 else
                          O.Filter(int(d["m"]), int(d["k"]), want[int(d["word_off"]): int(d["word_off"]) + O.words_for(int(d["m"]))]))
            assert secs[b] == O.encode_filter_section(fl)
        used = m.device_calls() - before
        assert (used > 0).sum() == n_dev, used                   # every entry of the context took its part
        # a layout whose word offsets do not ascend with the filter index stays on one device, and is still right
        perm = plan.desc.copy()
        order = np.argsort(-plan.desc["word_off"].astype(np.int64), kind="stable")
        cursor = 0
        for f in order:
            if int(perm[f]["m"]) == 0:
                continue
            perm[f]["word_off"] = cursor
            cursor += ((int(perm[f]["m"]) + 63) // 64 + 1) // 2 * 2
        back = m.build(plan.blob, plan.off, plan.fstart, perm, max(cursor, 2))
        for f in range(len(perm)):
            nw = O.words_for(int(perm[f]["m"]))
            if nw:
                assert np.array_equal(back[int(perm[f]["word_off"]): int(perm[f]["word_off"]) + nw],
                                      want[int(plan.desc[f]["word_off"]): int(plan.desc[f]["word_off"]) + nw])


def oracle_sets(rows):
    sets = (set(), set(), set())
    for r in rows:
        W.index_row(r, sets)
    return sets


def host_sets(rows):
    s = Hst.EntrySets()
    for r in rows:
        try:
            s.index_row(r)
        except Hst.HostError:
            pass
    return s.as_python_sets()


@pytest.mark.parametrize("n_dev", [2, 3, 8])
def test_ingest_counts_bitsets_sections_and_resident_arenas_identical(ctx, n_dev):
    """11 sets of uneven size, two files (parents) whose children straddle the parts, rows the host walker must finish in
    several parts: exact counts, statuses, fallback lists, bitsets, section bytes and the resident arenas' probe results
    equal the single-device context's and the oracle's."""
    from tests.test_ingest_gpu import ROWS_HOST, ROWS_UTF8_DEVICE
    sizes = [300, 5, 900, 0, 450, 1200, 64, 700, 1, 333, 800]
    row_sets, r = [], 0
    for i, n in enumerate(sizes):
        rows = synth.rows_json(r, n)
        r += n
        if i in (1, 4, 9):
            rows = rows + ROWS_HOST[:5] + ROWS_UTF8_DEVICE[:3]
        row_sets.append(rows)
    parent_of = [0, 0, 0, 1, 1, 0xFFFFFFFF, 1, 0, 1, 0, 1]
    single = I.device_ingest(ctx, row_sets, FPR, parent_of_set=parent_of, n_parents=2)
    with sharded(n_dev) as m:
        before = m.device_calls()
        multi = I.device_ingest(m, row_sets, FPR, parent_of_set=parent_of, n_parents=2)
        assert np.array_equal(multi.counts, single.counts)
        assert np.array_equal(multi.status, single.status) and not multi.status.any()
        assert np.array_equal(multi.fallback_rows, single.fallback_rows) and len(multi.fallback_rows) >= 15
        assert np.array_equal(multi.desc, single.desc)
        assert np.array_equal(multi.words, single.words)
        assert multi.stats.n_rows == sum(len(x) for x in row_sets)
        assert ((m.device_calls() - before) > 0).sum() >= 2          # (contiguous runs of about equal bytes: uneven sets can leave fewer parts than devices)
        # against the oracle: every set, and the two files' unions
        unions = [(set(), set(), set()), (set(), set(), set())]
        for s, rows in enumerate(row_sets):
            sets = host_sets(rows)
            for kind in range(3):
                assert int(multi.counts[s, kind]) == len(sets[kind]), (s, kind)
                want = O.build_sized(sorted(sets[kind]), FPR)
                assert np.array_equal(multi.filter_words(s, kind), want.words), (s, kind)
            if parent_of[s] != 0xFFFFFFFF:
                for u, x in zip(unions[parent_of[s]], sets):
                    u |= x
        for p in range(2):
            for kind in range(3):
                assert int(multi.counts[len(sizes) + p, kind]) == len(unions[p][kind])
                want = O.build_sized(sorted(unions[p][kind]), FPR)
                assert np.array_equal(multi.filter_words(len(sizes) + p, kind), want.words), ("file", p, kind)

        # the sections route, with the filters left resident as probe arenas
        def sections_of(c):
            rows = [x for rs in row_sets for x in rs]
            first = np.zeros(len(row_sets) + 1, dtype=np.uint32)
            first[1:] = np.cumsum([len(rs) for rs in row_sets])
            ing = c.ingest_rows(rows, first, parent_of, 2)
            fb = c.ingest_fallback_rows(ing)
            set_of_row = np.repeat(np.arange(len(row_sets)), np.diff(first.astype(np.int64)))
            entries, sets, kinds = I.host_walk_entries(rows, fb, set_of_row)
            c.ingest_add_entries(ing, entries, sets, kinds)
            counts, _ = c.ingest_finish(ing, len(row_sets) + 2)
            desc, _ = I.plan_desc(counts, FPR)
            secs, a_sets, a_par = c.ingest_build_sections(ing, desc, arenas=True)
            c.ingest_free(ing)
            return secs, a_sets, a_par, desc
        s_secs, s_sets, s_par, desc = sections_of(ctx)
        m_secs, m_sets, m_par, _ = sections_of(m)
        assert m_secs == s_secs
        exprs = [Q.FieldToken("level", "error"), Q.Token("timeout"), Q.And(Q.Field("nested.az"), Q.Token("payment")), Q.Token("nope"),
                 Q.Or(Q.FieldToken("service", "auth"), Q.Token("zzz")), None]
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = H.gpu_terms(ctx, cb)
        for (ma, sa, nb) in ((m_sets, s_sets, len(sizes)), (m_par, s_par, 2)):
            got = m.probe(ma, nb, terms, ops, poff)
            assert np.array_equal(got, ctx.probe(sa, nb, terms, ops, poff))
            reloaded, status = ctx.arena_load_sections(m_secs[:nb] if nb == len(sizes) else m_secs[len(sizes):])
            assert not status.any() and np.array_equal(got, ctx.probe(reloaded, nb, terms, ops, poff))
            ctx.arena_free(reloaded)
            m.arena_free(ma)
            ctx.arena_free(sa)


@pytest.mark.parametrize("n_dev", [2, 8])
def test_match_rows_identical_on_a_sharded_context(ctx, n_dev):
    from tests.test_ingest_gpu import ROWS_HOST
    rows = synth.rows_json(0, 5000)
    for i in (10, 700, 2049, 4999):
        rows[i] = ROWS_HOST[i % len(ROWS_HOST)]
    exprs = [Q.And(Q.FieldToken("level", "error"), Q.Field("nested.az")), Q.Or(Q.Token("timeout"), Q.FieldToken("service", "auth")), Q.Token("nope"), None]
    with sharded(n_dev) as m:
        before = m.device_calls()
        for e in exprs:
            cm = Q.CompiledMatcher(e)
            hits_s, fb_s = ctx.match_rows(rows, cm)
            hits_m, fb_m = m.match_rows(rows, cm)
            assert np.array_equal(hits_m, hits_s) and np.array_equal(fb_m, fb_s)
            for r in list(range(0, 5000, 397)) + [63, 64, 65]:
                if r not in fb_s:
                    assert bool(hits_m[r]) == Hst.match_row(e, rows[r]), (r, e)
        assert ((m.device_calls() - before) > 0).sum() == n_dev


def test_independent_callers_spread_over_the_devices(ctx):
    """Default thresholds: a flush of a few hundred rows stays on ONE device — and two flush workers running side by side
    land on different devices of a 2-entry context (each takes the device nobody holds), with results equal to the
    single-device context's."""
    row_sets = [synth.rows_json(b * 400, 400) for b in range(3)]
    want = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0, 0, 0], n_parents=1)
    with Context(device_ids(2)) as m:
        results, errors = {}, []

        def worker(k):
            try:
                for it in range(6):
                    results[(k, it)] = I.device_ingest(m, row_sets, FPR, parent_of_set=[0, 0, 0], n_parents=1)
            except Exception as exc:   # noqa: BLE001
                errors.append(exc)
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for res in results.values():
            assert np.array_equal(res.counts, want.counts) and np.array_equal(res.words, want.words)
        calls = m.device_calls()
        assert calls.min() >= 3 and calls.sum() == 12, calls         # 12 small ingests, neither device left idle
