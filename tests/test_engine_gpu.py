"""End-to-end tests of the engine mirror (IngestRows / Flush / Query / Merge) on the GPU, written to
read like the reference's own engine tests: same rows, same queries, same expected outcomes.
Bloom arithmetic runs in the HIP kernels (bsg_build / bsg_probe); the oracle is only the checker.

Scenarios after: query_cursor_test.go:442-524 (TestBlockStatsAccuracy), file_format_test.go:28-94
(TestMeasuredFilterSizing), :940-1055 (TestMergeRebuildsFilters), no_false_negatives_test.go:103-321 and
:467-611 (TestPropertyNoFalseNegatives), bloom_tree_engine_test.go:1867-1901 (file-level prune).
"""
import json

import numpy as np
import pytest

from bloomsearch_amd import host as Hst, query as Q
from oracle import oracle as O
from oracle import walker_oracle as W
from tests.test_host_tables import KEYS, _random_value, go_marshal

pytestmark = pytest.mark.gpu


DEVICE_INGEST = False
DEVICE_MATCH = False


@pytest.fixture(autouse=True, params=[(False, False), (True, False), (True, True)], ids=["host", "device-ingest", "device-ingest+match"])
def ingest_mode(request):
    """Every scenario runs three ways: indexRow and matchRowBytes on the host (the reference's order of work); DeviceIngest
    (rows walked / tokenized / deduplicated / counted by k_ingest_rows at flush and merge time); and DeviceMatch on top
    (the final row test of the surviving blocks by k_match_rows)."""
    global DEVICE_INGEST, DEVICE_MATCH
    DEVICE_INGEST, DEVICE_MATCH = request.param
    yield


def new_engine(ctx, **cfg):
    cfg.setdefault("DeviceIngest", DEVICE_INGEST)
    cfg.setdefault("DeviceMatch", DEVICE_MATCH)
    return Hst.Engine(ctx, **cfg)


def oracle_section(rows, fpr):
    """What the reference stores for these rows: indexRow over every row (ingest.go:55-89), one right-sized filter per
    kind (buildSizedBloomFilter, ingest.go:127-145), encodeFilterSection (file_format.go:343-384) — all by the oracle."""
    sets = (set(), set(), set())
    for r in rows:
        W.index_row(r, sets)
    return O.encode_filter_section([O.build_sized(sorted(s), fpr) for s in sets]), sets


def assert_file_equals_oracle(e, file_index, rows_of_partition, fpr):
    """Every block section and the file-level section of file `file_index` == the oracle's encode of the oracle's build of
    the oracle's entry sets of the rows that block / file holds (merge.go:516,771 and flush.go:204,253: rebuilt, right-sized)."""
    d = e.describe()["files"][file_index]
    all_rows = []
    for b, blk in enumerate(d["blocks"]):
        rows = rows_of_partition[blk["PartitionID"]]
        assert blk["Rows"] == len(rows)
        want, sets = oracle_section(rows, fpr)
        assert e.section_bytes(file_index, b) == want, "block %d (partition %r) differs from the oracle" % (b, blk["PartitionID"])
        assert blk["BloomEntryCounts"] == {"Fields": len(sets[0]), "Tokens": len(sets[1]), "FieldTokens": len(sets[2])}
        all_rows += rows
    want, sets = oracle_section(all_rows, fpr)
    assert e.section_bytes(file_index, -1) == want, "file-level section differs from the oracle"
    assert d["BloomEntryCounts"] == {"Fields": len(sets[0]), "Tokens": len(sets[1]), "FieldTokens": len(sets[2])}


def ingest_and_flush(engine, rows):
    engine.ingest_rows([go_marshal(r) for r in rows])
    engine.flush()


def result_ids(res):
    return {r["id"] for r in res["rows"]}


def test_block_stats_accuracy(ctx):
    e = new_engine(ctx, PartitionField="partition", BloomFalsePositiveRate=1e-6)
    ingest_and_flush(e, [
        {"id": 1.0, "partition": "a", "message": "alphaonly"},
        {"id": 2.0, "partition": "a", "message": "alphaonly"},
        {"id": 3.0, "partition": "b", "message": "betaonly"},
        {"id": 4.0, "partition": "b", "message": "betaonly"},
        {"id": 5.0, "partition": "b", "message": "betaonly"},
    ])
    res = e.query(Q.Token("alphaonly"))
    assert len(res["rows"]) == 2
    stats = res["stats"]["BlockStats"]
    assert len(stats) == 2
    skipped = [b for b in stats if b["BloomFilterSkipped"]]
    scanned = [b for b in stats if not b["BloomFilterSkipped"]]
    assert len(skipped) == 1 and len(scanned) == 1
    assert skipped[0]["RowsProcessed"] == 0 and skipped[0]["BytesProcessed"] == 0
    assert skipped[0]["TotalRows"] == 3 and skipped[0]["TotalBytes"] > 0
    assert scanned[0]["RowsProcessed"] == 2 == scanned[0]["TotalRows"] and scanned[0]["BytesProcessed"] > 0
    # BlockStats.Duration: the batched probe has one wall time for all blocks; every candidate block gets its share and none
    # reports zero (query_exec.go:578,598-600; asserted > 0 at query_handles_test.go:1062)
    assert all(b["Duration"] > 0 for b in stats)
    assert scanned[0]["Duration"] > skipped[0]["Duration"]      # the scanned block also carries the scan's time


def test_regex_query_is_pruned_by_its_field_guard(ctx):
    """Query.Regex: files / blocks are pruned by AndBloomQueries(bloom, RegexFieldGuardBloomQuery(regex)) (query_exec.go:220) —
    a block none of whose rows has the field cannot match FieldRegex on it — and rows are matched by bloom AND regex
    (row_matcher.go:353-368); with the device matcher on, the patterns only run on rows the guard's Field conditions keep."""
    e = new_engine(ctx, PartitionField="partition", BloomFalsePositiveRate=1e-6)
    ingest_and_flush(e, [
        {"id": 1.0, "partition": "a", "message": "timeout talking to db", "level": "error"},
        {"id": 2.0, "partition": "a", "message": "retry scheduled", "level": "info"},
        {"id": 3.0, "partition": "b", "note": "timeout in a field the regex does not name"},
        {"id": 4.0, "partition": "b", "note": "nothing"},
        {"id": 5.0, "partition": "c", "message": "all good", "level": "error", "svc": {"name": "payments"}},
    ])
    regex = Q.RegexAnd(Q.FieldRegex("message", "timeout|retry"), Q.FieldRegex("level", "^err"))
    res = e.query(None, regex)
    assert result_ids(res) == {1.0}
    stats = {b["BlockOffset"]: b for b in res["stats"]["BlockStats"]}
    assert sum(b["BloomFilterSkipped"] for b in stats.values()) == 1       # partition b has neither message nor level
    # bloom AND regex; regex alone beneath a path; a guard that keeps everything
    assert result_ids(e.query(Q.Token("retry"), Q.FieldRegex("level", "^(info|error)$"))) == {2.0}
    assert result_ids(e.query(None, Q.FieldRegex("svc", "^pay"))) == {5.0}
    assert result_ids(e.query(None, Q.RegexOr(Q.FieldRegex("note", "^nothing$"), Q.FieldRegex("id", "^1\\.0$")))) == {1.0, 4.0}
    assert result_ids(e.query(Q.Field("partition"), {"ExpressionType": "CONDITION", "Condition": None})) == {1.0, 2.0, 3.0, 4.0, 5.0}


def test_measured_filter_sizing(ctx):
    e = new_engine(ctx)
    ingest_and_flush(e, [{"id": "row%d" % i, "color": "red"} for i in range(100)])
    d = e.describe()
    assert len(d["files"]) == 1 and len(d["files"][0]["blocks"]) == 1
    want = {"Fields": 2, "Tokens": 101, "FieldTokens": 101}
    f = d["files"][0]
    assert f["BloomEntryCounts"] == want and f["blocks"][0]["BloomEntryCounts"] == want
    for filt in (f["filters"], f["blocks"][0]["filters"]):
        for got, n in zip(filt, (2, 101, 101)):
            m, k = O.estimate_parameters(n, 0.001)      # == bloom.NewWithEstimates(count, fpr).Cap()/K()
            assert (got["m"], got["k"]) == (m, k)
    assert f["blocks"][0]["filters"][0]["m"] != O.estimate_parameters(100, 0.001)[0]   # not row-count sized
    # size identity of the stored section: 1 + sum(4 + 24 + 8*ceil(m/64)) + 4
    assert f["blocks"][0]["BloomFilterSize"] == 1 + sum(4 + 24 + 8 * O.words_for(x["m"]) for x in f["blocks"][0]["filters"]) + 4


def test_engine_filter_bytes_equal_oracle_build(ctx):
    """The wire bytes the engine stores == encodeFilterSection of oracle-built filters over the same rows."""
    rows = [go_marshal({"id": i, "msg": "hello world %d" % (i % 7), "user": {"name": "U%d" % (i % 13)}}) for i in range(500)]
    e = new_engine(ctx, MaxBufferedRows=100000)
    e.ingest_rows(rows)
    e.flush()
    sets = (set(), set(), set())
    for r in rows:
        W.index_row(r, sets)
    want = O.encode_filter_section([O.build_sized(sorted(s), 0.001) for s in sets])
    assert e.section_bytes(0, 0) == want
    assert e.section_bytes(0, -1) == want        # single block: file-level union == block sets


def test_no_false_negative_regressions(ctx):
    e = new_engine(ctx)
    ingest_and_flush(e, [{"id": 1, "user_id": 1234567}, {"id": 2, "big": 9007199254740993}])
    assert result_ids(e.query(Q.FieldToken("user_id", "1234567"))) == {1}
    assert result_ids(e.query(Q.FieldToken("big", "9007199254740993"))) == {2}
    assert result_ids(e.query(Q.Token("1234567"))) == {1}

    e = new_engine(ctx)
    ingest_and_flush(e, [{"id": 1, "a.b": "hello"}, {"id": 2, "a": {"b": "world"}}, {"id": 3, "user.name": "x"}, {"id": 4, ".a": "xyz"}])
    assert result_ids(e.query(Q.Field("a.b"))) == {1, 2}
    assert result_ids(e.query(Q.FieldToken("a.b", "hello"))) == {1}
    assert result_ids(e.query(Q.FieldToken("a.b", "world"))) == {2}
    assert result_ids(e.query(Q.Field("a"))) == {1, 2}
    assert result_ids(e.query(Q.Field("user"))) == {3}      # the FieldRegex("user", ...) bloom guard's Field term

    e = new_engine(ctx)
    ingest_and_flush(e, [{"id": 1, "ab": "x"}, {"id": 2, "a*": 1}, {"id": 3, "back\\slash": "v", "q?x": "y"}])
    assert result_ids(e.query(Q.Field("a*"))) == {2}
    assert result_ids(e.query(Q.FieldToken("a*", "1"))) == {2}
    assert result_ids(e.query(Q.Field("back\\slash"))) == {3}
    assert result_ids(e.query(Q.FieldToken("q?x", "y"))) == {3}

    e = new_engine(ctx)
    ingest_and_flush(e, [{"id": 1, "user": {"name": "x"}}, {"id": 2, "other": "y"}])
    assert result_ids(e.query(Q.Field("user"))) == {1}

    e = new_engine(ctx)
    ingest_and_flush(e, [{"id": 1, "p": {"x": 7, "y": "hi"}}, {"id": 2, "ts": "2020-01-02T03:04:05Z"}, {"id": 3, "data": "aGk="}, {"id": 4, "n": None}])
    assert result_ids(e.query(Q.FieldToken("p.x", "7"))) == {1}
    assert result_ids(e.query(Q.FieldToken("p.y", "hi"))) == {1}
    assert result_ids(e.query(Q.FieldToken("ts", "2020-01-02t03:04:05z"))) == {2}
    assert result_ids(e.query(Q.FieldToken("data", "agk="))) == {3}
    assert result_ids(e.query(Q.Field("n"))) == {4}
    assert result_ids(e.query(Q.FieldToken("n", "null"))) == set()


def test_file_level_prune_produces_no_block_stats(ctx):
    # bloom_tree_engine_test.go:1867-1901: file 1 has fields {id, service}; Field("message") prunes it at FILE level
    e = new_engine(ctx, BloomFalsePositiveRate=0.01)
    ingest_and_flush(e, [{"id": 1, "service": "auth"}])
    ingest_and_flush(e, [{"id": 2, "service": "auth", "message": "boom"}])
    res = e.query(Q.Field("message"))
    assert result_ids(res) == {2}
    assert len(res["stats"]["BlockStats"]) == 1
    assert res["stats"]["FilesConsidered"] == 2 and res["stats"]["FilesBloomSkipped"] == 1
    # nil query: no bloom conditions => nothing read, everything scanned (query_exec.go:503-508)
    res = e.query(None)
    assert result_ids(res) == {1, 2} and not any(b["BloomFilterSkipped"] for b in res["stats"]["BlockStats"])


def test_property_no_false_negatives(ctx):
    rng = np.random.default_rng(7)
    rows = []
    for i in range(100):
        obj = {KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}
        obj["id"] = i
        rows.append(obj)
    e = new_engine(ctx, PartitionField="id", MaxBufferedRows=100000)    # one block per row: pruning really bites
    ingest_and_flush(e, rows)
    checked = 0
    for obj in rows[::3]:
        raw = go_marshal(obj)
        fields, tokens, fts = W.index_row(raw)
        for q in [Q.Field(f) for f in sorted(fields)[:4]] + [Q.Token(t) for t in sorted(tokens)[:4]] + \
                 [Q.FieldToken(*ft.split("::", 1)) for ft in sorted(fts)[:4] if ft.count("::") == 1]:
            res = e.query(q)
            assert obj["id"] in result_ids(res), (q, raw)
            for r in res["rows"]:                       # and nothing that does not match comes back
                assert W.matches_bloom_expression(go_marshal(r), q)
            checked += 1
    assert checked > 100


def test_surviving_block_sets_match_oracle(ctx):
    """BloomFilterSkipped per block == oracle probe over the engine's own stored filter bytes."""
    rows = [{"id": i, "partition": "p%d" % (i % 9), "msg": "w%d shared" % (i % 23), "k%d" % (i % 5): i} for i in range(300)]
    e = new_engine(ctx, PartitionField="partition", MaxBufferedRows=100000)
    ingest_and_flush(e, rows)
    n_blocks = len(e.describe()["files"][0]["blocks"])
    filters = [O.parse_filter_section(e.section_bytes(0, b)) for b in range(n_blocks)]
    for q in (Q.Token("w3"), Q.And(Q.Field("k2"), Q.Token("shared")), Q.Or(Q.FieldToken("msg", "w22"), Q.Field("k4")), Q.Token("zzz")):
        res = e.query(q)
        got = [not b["BloomFilterSkipped"] for b in res["stats"]["BlockStats"]]

        def ev(x, fl):
            et = x["ExpressionType"]
            if et == "CONDITION":
                kind, s = Q.term_of(x["Condition"])
                return fl[kind].test(s)
            vals = [ev(c, fl) for c in x["Children"]]
            return all(vals) if et == "AND" else any(vals)
        want = [ev(q, fl) for fl in filters]
        if sum(want) == 0 and res["stats"]["FilesBloomSkipped"] == 1:
            assert got == []
        else:
            assert got == want


def test_merge_rebuilds_right_sized_filters(ctx):
    # file_format_test.go:940-1055: merged file's filters are rebuilt for the UNION's distinct counts (never OR-ed)
    e = new_engine(ctx)
    small, large = [{"id": "a%d" % i, "kind": "x"} for i in range(20)], [{"id": "b%d" % i, "kind": "x"} for i in range(200)]
    ingest_and_flush(e, small)
    ingest_and_flush(e, large)
    assert len(e.describe()["files"]) == 2
    pid = e.describe()["files"][0]["blocks"][0]["PartitionID"]
    assert_file_equals_oracle(e, 0, {pid: [go_marshal(r) for r in small]}, 0.001)      # the flushed files first (flush.go:204,253)
    assert_file_equals_oracle(e, 1, {pid: [go_marshal(r) for r in large]}, 0.001)
    e.merge()
    d = e.describe()
    assert len(d["files"]) == 1 and len(d["files"][0]["blocks"]) == 1
    # the merged file against the ORACLE, not against another mode of the product: block and file sections are the oracle's
    # encode of the oracle's right-sized build of the union's entry sets (merge.go:516,771)
    assert_file_equals_oracle(e, 0, {pid: [go_marshal(r) for r in small + large]}, 0.001)
    want = {"Fields": 2, "Tokens": 221, "FieldTokens": 221}
    assert d["files"][0]["BloomEntryCounts"] == want and d["files"][0]["blocks"][0]["BloomEntryCounts"] == want
    assert d["files"][0]["filters"][1]["m"] == O.estimate_parameters(221, 0.001)[0]
    assert len(e.query(Q.Token("a7"))["rows"]) == 1 and len(e.query(Q.Token("b150"))["rows"]) == 1
    assert len(e.query(Q.Token("zzzabsent0"))["rows"]) == 0


def test_engine_errors(ctx):
    with pytest.raises(Hst.HostError) as ei:
        new_engine(ctx, BloomFalsePositiveRate=1.5)
    assert ei.value.code == -101                      # ErrInvalidConfig
    e = new_engine(ctx)
    with pytest.raises(Hst.HostError) as ei:          # whole batch rejected, nothing buffered (ingest.go:378-397)
        e.ingest_rows([b'{"id":1}', b'[1,2]'])
    assert ei.value.code == -103
    e.flush()
    assert e.describe()["files"] == []
    e.stop()
    with pytest.raises(Hst.HostError) as ei:
        e.ingest_rows([b'{"id":1}'])
    assert ei.value.code == -102                      # ErrEngineStopped
    assert e.query(None)["rows"] == []                # queries still served after Stop


def test_flush_triggers(ctx):
    e = new_engine(ctx, MaxRowGroupRows=10, MaxBufferedRows=1000)
    e.ingest_rows([go_marshal({"id": i}) for i in range(9)])
    assert e.describe()["files"] == []
    e.ingest_rows([go_marshal({"id": 9})])           # partition hit MaxRowGroupRows => flush (ingest.go:497-503)
    assert len(e.describe()["files"]) == 1
    e = new_engine(ctx, MaxBufferedRows=5, PartitionField="id")
    e.ingest_rows([go_marshal({"id": i}) for i in range(5)])     # buffer hit MaxBufferedRows (ingest.go:512-516)
    d = e.describe()
    assert len(d["files"]) == 1 and len(d["files"][0]["blocks"]) == 5


def test_corrupt_block_section_is_isolated(ctx):
    """A block whose stored filter section fails its CRC32C (checked on the device) is neither pruned nor scanned:
    it gets a totals-only stats entry and the failure surfaces in the result's errors; every other block behaves
    as before (query_exec.go:580-590, recordUnreadBlocks :625-639)."""
    rows = [{"id": i, "partition": "p%d" % (i % 4), "msg": "only%d common" % (i % 4)} for i in range(40)]
    e = new_engine(ctx, PartitionField="partition", MaxBufferedRows=100000, BloomFalsePositiveRate=1e-6)
    ingest_and_flush(e, rows)
    clean = e.query(Q.Token("only2"))
    assert len(clean["rows"]) == 10 and clean["stats"]["Errors"] == []
    assert sum(b["BloomFilterSkipped"] for b in clean["stats"]["BlockStats"]) == 3
    e.corrupt_section_byte(0, 2, 40)           # the block that holds "only2"
    res = e.query(Q.Token("only2"))
    assert len(res["stats"]["Errors"]) == 1 and "invalid hash" in res["stats"]["Errors"][0]
    assert res["rows"] == []                    # its rows are not scanned
    stats = res["stats"]["BlockStats"]
    assert len(stats) == 4
    unread = [b for b in stats if not b["BloomFilterSkipped"] and b["RowsProcessed"] == 0]
    assert len(unread) == 1 and unread[0]["TotalRows"] == 10
    assert sum(b["BloomFilterSkipped"] for b in stats) == 3
    other = e.query(Q.Token("only1"))           # other blocks still answer normally
    assert len(other["rows"]) == 10 and len(other["stats"]["Errors"]) == 1


def test_c1_config_100k_rows_fieldtoken_level_error(ctx):
    """BASELINE configs[0] (the reference's own CPU-runnable case) through the engine mirror: 100 000 synthetic log
    rows, 10 flushes of 10 000 rows (10 files x 1 block), FieldToken("level", "error").  Every block holds the term,
    so nothing is pruned; the delivered row set is exactly the rows whose level is "error" (~25 %)."""
    import time
    from bloomsearch_amd import synth
    e = new_engine(ctx, MaxRowGroupRows=10000, MaxBufferedRows=10_000_000, MaxBufferedBytes=1 << 40)
    t_ing = 0.0
    for f in range(10):
        rows = synth.rows_json(f * 10000, 10000)
        t0 = time.perf_counter()
        e.ingest_rows(rows)                     # 10 000th row of the partition triggers the flush
        t_ing += time.perf_counter() - t0
    d = e.describe()
    assert len(d["files"]) == 10 and all(len(f["blocks"]) == 1 and f["blocks"][0]["Rows"] == 10000 for f in d["files"])
    c = d["files"][0]["blocks"][0]["BloomEntryCounts"]
    assert c["Fields"] == 9 and 19000 < c["Tokens"] < 20000 and 19000 < c["FieldTokens"] < 20000
    res = e.query(Q.FieldToken("level", "error"))
    want = int((synth.draws(0, 100000)["level"] == synth.LEVELS.index("error")).sum())
    assert len(res["rows"]) == want and 0.24 < want / 100000 < 0.26
    assert all(r["level"] == "error" for r in res["rows"])
    st = res["stats"]["BlockStats"]
    assert len(st) == 10 and not any(b["BloomFilterSkipped"] for b in st) and sum(b["RowsProcessed"] for b in st) == 100000
    miss = e.query(Q.FieldToken("level", "fatal"))
    assert miss["rows"] == [] and miss["stats"]["FilesBloomSkipped"] == 10 and miss["stats"]["BlockStats"] == []
    print("ingest+flush (%s): %.2f us/row" % ("device ingest" if DEVICE_INGEST else "host walk/tokenize/dedup + GPU build",
                                              t_ing / 100000 * 1e6))


def test_device_ingest_writes_the_same_bytes_as_host_ingest(ctx):
    """Same rows through both ingest modes, flush + second flush + merge: every stored filter section (block-level and
    file-level) and every BloomEntryCounts stamp is byte-identical — the device walker changes where the work runs,
    not one bit of what is written."""
    rng = np.random.default_rng(23)
    batches = []
    for b in range(2):
        rows = []
        for i in range(400):
            obj = {KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}
            obj["partition"] = "p%d" % (i % 4)
            obj["Msg"] = "Shared WORDS %d and MiXeD case" % (i % 17)
            rows.append(go_marshal(obj))
        batches.append(rows)
    engines = [Hst.Engine(ctx, PartitionField="partition", MaxBufferedRows=100000, DeviceIngest=mode) for mode in (False, True)]
    for e in engines:
        for rows in batches:
            e.ingest_rows(rows)
            e.flush()

    def snapshot(e):
        d = e.describe()
        return d, [[e.section_bytes(f, b) for b in range(-1, len(fl["blocks"]))] for f, fl in enumerate(d["files"])]
    (d0, s0), (d1, s1) = snapshot(engines[0]), snapshot(engines[1])
    assert d0 == d1 and s0 == s1 and len(d0["files"]) == 2 and len(d0["files"][0]["blocks"]) == 4
    for e in engines:
        e.merge()
    (d0, s0), (d1, s1) = snapshot(engines[0]), snapshot(engines[1])
    assert d0 == d1 and s0 == s1 and len(d0["files"]) == 1
    # ... and both equal the oracle: four merged blocks (one per partition, the rows of both flushes) and the file-level
    # filters rebuilt at the union's size (merge.go:706-804, :516)
    by_partition = {}
    for rows in batches:
        for i, r in enumerate(rows):
            by_partition.setdefault("p%d" % (i % 4), []).append(r)
    for e in engines:
        assert_file_equals_oracle(e, 0, by_partition, 0.001)


def test_queries_are_the_same_whatever_the_arena_budget_keeps_resident(ctx):
    """The engine mirror leases its files' block filters from the library's resident-arena cache (csrc/cache_api.inc).  Five files,
    the same queries under a budget that holds everything, one that holds about two files (arenas are evicted between and inside
    queries and decoded again from the stored sections) and one that holds nothing (every arena serves its query and is freed):
    identical rows and BlockStats; a corrupted section is re-read on every query and recovers when the bytes do (only clean decodes
    become resident); merged-away files are forgotten."""
    e = new_engine(ctx, PartitionField="partition", MaxBufferedRows=100000)
    for f in range(5):
        ingest_and_flush(e, [{"id": f * 1000 + i, "partition": "p%d" % (i % 6), "msg": "w%d f%d shared" % (i % 17, f), "k%d" % (i % 4): i} for i in range(240)])
    queries = [Q.Token("w3"), Q.And(Q.Field("k2"), Q.Token("f4")), Q.Or(Q.FieldToken("msg", "w16"), Q.Token("f0")), Q.Token("zzz"), Q.Token("shared")]

    def answers():
        out = []
        for q in queries:
            r = e.query(q)
            out.append((sorted(json.dumps(x, sort_keys=True) for x in r["rows"]), [(b["FileID"], b["BlockOffset"], b["BloomFilterSkipped"], b["RowsProcessed"]) for b in r["stats"]["BlockStats"]], r["stats"]["Errors"]))
        return out
    try:
        ctx.set_arena_budget(1 << 40)
        ctx.arena_cache_stats(reset=True)
        want = answers()
        st = ctx.arena_cache_stats()
        assert st["resident_files"] == 5 and st["leases"] == 0 and (DEVICE_INGEST or st["misses"] == 5) and st["hits"] >= 10     # (file-level filters prune "zzz" everywhere and "f4" in four files: those never lease)
        one = st["resident_bytes"] // 5
        ctx.set_arena_budget(2 * one + one // 2)
        assert ctx.arena_cache_stats()["resident_files"] == 2
        ctx.arena_cache_stats(reset=True)
        assert answers() == want
        st = ctx.arena_cache_stats()
        assert st["evictions"] > 0 and st["resident_bytes"] <= st["budget_bytes"] and st["leases"] == 0
        ctx.set_arena_budget(0)
        ctx.arena_cache_stats(reset=True)
        assert answers() == want
        st = ctx.arena_cache_stats()
        assert st["resident_files"] == 0 and st["rejected_over_budget"] > 0 and st["hits"] == 0 and st["leases"] == 0 and st["leased_dead_bytes"] == 0
        # a corrupt section: the block is unread in every query while the bytes are bad (never cached), the other files' arenas stay resident
        ctx.set_arena_budget(1 << 40)
        answers()
        e.corrupt_section_byte(2, 1, 40)
        bad1, bad2 = e.query(Q.Token("shared")), e.query(Q.Token("shared"))
        assert len(bad1["stats"]["Errors"]) == 1 and bad1["stats"]["Errors"] == bad2["stats"]["Errors"]
        assert ctx.arena_cache_stats()["resident_files"] == 4 and ctx.arena_cache_stats()["rejected_dirty"] >= 2
        e.corrupt_section_byte(2, 1, 40)           # the same flip again: the bytes are good again, the next query recovers
        assert e.query(Q.Token("shared"))["stats"]["Errors"] == [] and ctx.arena_cache_stats()["resident_files"] == 5
        e.merge()                                   # five sources tombstoned, one merged file
        st = ctx.arena_cache_stats()
        assert st["resident_files"] <= 1 and st["forgotten"] >= 5
        merged = e.query(Q.Token("w3"))
        assert sorted(json.dumps(x, sort_keys=True) for x in merged["rows"]) == want[0][0]
    finally:
        ctx.set_arena_budget(32 << 30)
