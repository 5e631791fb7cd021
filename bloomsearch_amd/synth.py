"""Deterministic synthetic workload: log-shaped rows after the
reference's benchRows (bench_test.go:24-49), driven by a counter-based
SplitMix64 (seed 0xB100F5EA4C4) so any row range can be produced independently,
either as marshaled JSON rows (what IngestRows sees) or directly as the
distinct bloom entry sets of a block (what buildFilters sees) without
materialising the rows.  tests/test_synth.py checks the two agree.

Row r:  {"level", "message" (12 words), "nested": {"az", "region"}, "service",
         "tags" [2 words], "timestamp": 1700000000 + r, "user_id" < 100000}
(keys in Go json.Marshal's sorted-map order).
"""
from __future__ import annotations

import numpy as np

SEED = 0xB100F5EA4C4
LEVELS = ["debug", "info", "warn", "error"]
SERVICES = ["auth", "payment", "search", "gateway", "billing"]
WORDS = ["connection", "timeout", "retry", "database", "request", "processed",
         "failed", "succeeded", "cache", "miss", "upstream", "latency", "shard"]
N_REGIONS, N_AZS, N_USERS, MSG_WORDS = 8, 3, 100000, 12
TS_BASE = 1700000000
DRAWS_PER_ROW = 32  # counter stride per row (19 used)
FIELD_PATHS = ["level", "message", "nested", "nested.az", "nested.region", "service", "tags", "timestamp", "user_id"]


def splitmix64(counter: np.ndarray, seed: int = SEED) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (np.asarray(counter, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def draws(r0: int, n: int, seed: int = SEED) -> dict:
    """All random draws of rows [r0, r0+n)."""
    r = (np.arange(r0, r0 + n, dtype=np.uint64) * np.uint64(DRAWS_PER_ROW))[:, None]
    z = splitmix64(r + np.arange(19, dtype=np.uint64)[None, :], seed)
    return {
        "level": (z[:, 0] % np.uint64(len(LEVELS))).astype(np.int64),
        "service": (z[:, 1] % np.uint64(len(SERVICES))).astype(np.int64),
        "words": (z[:, 2:14] % np.uint64(len(WORDS))).astype(np.int64),
        "user_id": (z[:, 14] % np.uint64(N_USERS)).astype(np.int64),
        "region": (z[:, 15] % np.uint64(N_REGIONS)).astype(np.int64),
        "az": (z[:, 16] % np.uint64(N_AZS)).astype(np.int64),
        "tags": (z[:, 17:19] % np.uint64(len(WORDS))).astype(np.int64),
    }


def rows_json(r0: int, n: int, seed: int = SEED) -> list[bytes]:
    """Rows as Go's json.Marshal(map[string]any) would emit them (sorted keys, no spaces)."""
    d = draws(r0, n, seed)
    out = []
    for i in range(n):
        msg = " ".join(WORDS[w] for w in d["words"][i])
        out.append((
            '{"level":"%s","message":"%s","nested":{"az":"az-%d","region":"region-%d"},'
            '"service":"%s","tags":["%s","%s"],"timestamp":%d,"user_id":%d}'
            % (LEVELS[d["level"][i]], msg, d["az"][i], d["region"][i], SERVICES[d["service"][i]],
               WORDS[d["tags"][i][0]], WORDS[d["tags"][i][1]], TS_BASE + r0 + i, d["user_id"][i])).encode())
    return out


def _fmt_uints(values: np.ndarray, prefix: bytes = b""):
    """Decimal ASCII of each value with a constant prefix -> (u8 blob, lengths), grouped by digit count."""
    values = np.asarray(values, dtype=np.uint64)
    blobs, lens = [], []
    nd = np.ones(len(values), dtype=np.int64)
    for e in range(1, 20):
        nd += values >= np.uint64(10 ** e)
    P = len(prefix)
    pre = np.frombuffer(prefix, dtype=np.uint8)
    for d in np.unique(nd):
        v = values[nd == d]
        mat = np.empty((len(v), P + int(d)), dtype=np.uint8)
        mat[:, :P] = pre
        pw = np.uint64(10) ** np.arange(int(d) - 1, -1, -1, dtype=np.uint64)
        mat[:, P:] = ((v[:, None] // pw[None, :]) % np.uint64(10)).astype(np.uint8) + 48
        blobs.append(mat.reshape(-1))
        lens.append(np.full(len(v), P + int(d), dtype=np.uint32))
    if not blobs:
        return np.zeros(0, np.uint8), np.zeros(0, np.uint32)
    return np.concatenate(blobs), np.concatenate(lens)


def _fixed(strings) -> tuple[np.ndarray, np.ndarray]:
    bs = [s.encode() for s in strings]
    return (np.frombuffer(b"".join(bs), dtype=np.uint8), np.asarray([len(b) for b in bs], dtype=np.uint32))


def block_entry_sets(r0: int, n: int, seed: int = SEED):
    """Distinct bloom entries of the block holding rows [r0, r0+n), as indexRow would
    collect them (ingest.go:55-102): -> [(blob, lengths)] * 3 for field / token / field::token."""
    d = draws(r0, n, seed)
    ts = np.arange(TS_BASE + r0, TS_BASE + r0 + n, dtype=np.uint64)
    uids = np.unique(d["user_id"]).astype(np.uint64)
    lv = [LEVELS[i] for i in np.unique(d["level"])]
    sv = [SERVICES[i] for i in np.unique(d["service"])]
    mw = [WORDS[i] for i in np.unique(d["words"])]
    tw = [WORDS[i] for i in np.unique(d["tags"])]
    rg = ["region-%d" % i for i in np.unique(d["region"])]
    az = ["az-%d" % i for i in np.unique(d["az"])]
    fields = _fixed(FIELD_PATHS)
    small_tokens = sorted(set(lv) | set(sv) | set(mw) | set(tw) | set(rg) | set(az))
    tok_parts = [_fmt_uints(ts), _fmt_uints(uids), _fixed(small_tokens)]
    small_ft = (["level::" + x for x in lv] + ["service::" + x for x in sv] + ["message::" + x for x in mw] +
                ["tags::" + x for x in tw] + ["nested.region::" + x for x in rg] + ["nested.az::" + x for x in az])
    ft_parts = [_fmt_uints(ts, b"timestamp::"), _fmt_uints(uids, b"user_id::"), _fixed(small_ft)]
    cat = lambda parts: (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    return [fields, cat(tok_parts), cat(ft_parts)]


def lengths_to_offsets(lengths: np.ndarray) -> np.ndarray:
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    np.cumsum(lengths, out=off[1:])
    assert off[-1] < 2 ** 32, "entry blob exceeds the u32 offset range of one bsg_build call"
    return off.astype(np.uint32)


def make_queries(n_queries, workload, seed):
    """C2 query batch.  'needle': And(FT(level), FT(service), FT(user_id)) — a log search for one
    user's events; 'lowcard': SURVEY C2's And(FT(level), FT(service), FT(nested.region)).
    Every position draws an absent value with probability 1/4."""
    from . import query as Q
    rng = np.random.default_rng(seed)
    exprs = []
    if workload == "c4":
        # BASELINE configs[3] / SURVEY C4: 8-term Or(FieldToken...) over the low-cardinality fields,
        # every position absent with probability 1/2 (an Or of present values alone would keep every block).
        def pick(field, present, absent):
            return Q.FieldToken(field, present() if rng.random() >= 0.5 else absent())
        for _ in range(n_queries):
            w = lambda: WORDS[rng.integers(0, len(WORDS))]
            nw = lambda: "absent-word-%d" % rng.integers(0, 8)
            exprs.append(Q.Or(
                pick("level", lambda: LEVELS[rng.integers(0, 4)], lambda: "absent-level-%d" % rng.integers(0, 4)),
                pick("service", lambda: SERVICES[rng.integers(0, 5)], lambda: "absent-svc-%d" % rng.integers(0, 4)),
                pick("nested.region", lambda: "region-%d" % rng.integers(0, 8), lambda: "region-%d" % rng.integers(8, 12)),
                pick("nested.az", lambda: "az-%d" % rng.integers(0, 3), lambda: "az-%d" % rng.integers(3, 6)),
                pick("tags", w, nw), pick("tags", w, nw), pick("message", w, nw), pick("message", w, nw)))
        return exprs
    for _ in range(n_queries):
        lv = LEVELS[rng.integers(0, 4)] if rng.random() >= 0.25 else "absent-level-%d" % rng.integers(0, 4)
        sv = SERVICES[rng.integers(0, 5)] if rng.random() >= 0.25 else "absent-svc-%d" % rng.integers(0, 4)
        if workload == "needle":
            third = Q.FieldToken("user_id", str(int(rng.integers(0, N_USERS * 4 // 3))))
        else:
            rg = "region-%d" % rng.integers(0, 8) if rng.random() >= 0.25 else "region-%d" % rng.integers(8, 12)
            third = Q.FieldToken("nested.region", rg)
        exprs.append(Q.And(Q.FieldToken("level", lv), Q.FieldToken("service", sv), third))
    return exprs
