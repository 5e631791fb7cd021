"""Seed sweep of the device walker / tokenizer / dedup / row matcher against the oracle (GPU box):
    python tools/fuzz_walker.py [first_seed] [n_seeds]
Three generators per seed (the tests' own, re-seeded): the reference's property generator (escapes, UTF-8), printable
ASCII with random spacing and nesting, multi-script Unicode with every white-space class; each ingested validated and
trusted, counts and bitsets compared with the oracle's build of the oracle's sets; the same rows DAMAGED (byte flips,
deletions, insertions, truncations) through the validating walk, compared with what the host walker alone keeps; then
random expressions through k_match_rows (intact and damaged rows) against the host matcher.  Exits non-zero on the first
difference."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bloomsearch_amd import host as Hst, ingest as I, query as Q
from bloomsearch_amd.gpu import Context
from oracle import oracle as O
from oracle import walker_oracle as W
from tests.test_host_tables import KEYS, _random_value, go_marshal

FPR = 0.001
POOLS = ["abcXYZ019-_.", "éñüßøåçœ", "ÀÉÜÑØÅ",
         "日本語中文한국어", "абвгд", "АБВГД",
         "αβγδ", "ΑΒΓΔ", "\U0001F600\U0001F389\U0001F680", "   　\t",
         "     ", "ǅǈǋ", "İıſK", "ⰯꟀꟐ\U00010570",
         "\"\\/\b\f\n\r\t<>&"]
ALPHA = [chr(c) for c in range(0x20, 0x7F) if chr(c) not in '"\\']


def gen_rows(rng, kind, n):
    rows = []
    if kind == 0:
        for _ in range(n):
            rows.append(go_marshal({KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}))
        return rows
    if kind == 1:
        def text(k):
            return "".join(ALPHA[rng.integers(0, len(ALPHA))] for _ in range(k))

        def val(depth):
            r = rng.random()
            if depth < 6 and r < 0.25:
                return {text(rng.integers(0, 9)): val(depth + 1) for _ in range(rng.integers(0, 4))}
            if depth < 6 and r < 0.4:
                return [val(depth + 1) for _ in range(rng.integers(0, 4))]
            if r < 0.5:
                return int(rng.integers(-10 ** 15, 10 ** 15))
            if r < 0.55:
                return float(rng.normal()) * 10 ** int(rng.integers(-20, 20))
            if r < 0.6:
                return [None, True, False][rng.integers(0, 3)]
            return text(rng.integers(0, 60))
        for _ in range(n):
            sep = [(",", ":"), (", ", ": "), (" , ", " : ")][rng.integers(0, 3)]
            rows.append(json.dumps({text(rng.integers(0, 14)): val(0) for _ in range(rng.integers(0, 6))}, separators=sep).encode())
        return rows

    def utext(k):
        out = []
        for _ in range(k):
            p = POOLS[rng.integers(0, len(POOLS))]
            out.append(p[rng.integers(0, len(p))])
        return "".join(out)
    for _ in range(n):
        obj = {utext(rng.integers(1, 7)): (utext(rng.integers(0, 30)) if rng.random() < 0.8 else [utext(3), int(rng.integers(0, 99)), {utext(2): utext(5)}])
               for _ in range(rng.integers(1, 5))}
        rows.append(json.dumps(obj, ensure_ascii=bool(rng.random() < 0.3), separators=(",", ":")).encode())
    return rows


def mutate(rng, rows):
    """byte flips, deletions, insertions and truncations of valid rows: what an attacker (or a torn write) hands the walker"""
    out = []
    for r in rows:
        b = bytearray(r)
        for _ in range(int(rng.integers(1, 4))):
            if not b:
                break
            p = int(rng.integers(0, len(b)))
            op = int(rng.integers(0, 4))
            if op == 0:
                b[p] = int(rng.integers(0, 256))
            elif op == 1:
                del b[p]
            elif op == 2:
                b.insert(p, int(rng.integers(0, 256)))
            else:
                del b[p:]
        out.append(bytes(b))
    return out


def host_sets(rows):
    """the C++ host walker's (lenient, Go-like) result: the reference for rows that are not valid JSON"""
    s = Hst.EntrySets()
    for r in rows:
        try:
            s.index_row(r)
        except Hst.HostError:
            pass
    out = []
    for kind in range(3):                      # raw bytes: a damaged key may not be UTF-8 any more (keys are copied undecoded)
        blob, ln = s.export(kind)
        off = np.concatenate([[0], np.cumsum(ln, dtype=np.int64)]).astype(np.int64)
        raw = blob.tobytes()
        out.append({raw[off[i]: off[i + 1]] for i in range(len(ln))})
    return tuple(out)


def oracle_sets(rows):
    sets = (set(), set(), set())
    for r in rows:
        W.index_row(r, sets)
    return sets


def check(res, idx, sets, what):
    for kind in range(3):
        if int(res.counts[idx, kind]) != len(sets[kind]):
            sys.exit("%s kind %d: device counts %d entries, oracle %d" % (what, kind, int(res.counts[idx, kind]), len(sets[kind])))
        want = O.build_sized(sorted(sets[kind]), FPR)
        d = res.desc[idx * 3 + kind]
        if (int(d["m"]), int(d["k"])) != (want.m, want.k) or not np.array_equal(res.filter_words(idx, kind), want.words):
            sys.exit("%s kind %d: bitset differs" % (what, kind))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    # odd seeds run on a 3-entry context that cuts every ingest and every match into one part per entry (lab key 8) and uploads
    # the rows in the smallest chunks the library takes: the row indices of hand-backs and verdicts must survive both cuts
    ctxs = [Context((0,)), Context((0, 0, 0))]
    ctxs[1].set_lab(8, 1)
    ctxs[1].set_ingest_chunk(1 << 16)
    n_rows = n_fb = n_match = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        ctx = ctxs[seed & 1]
        for kind in range(3):
            row_sets = [gen_rows(rng, kind, int(rng.integers(1, 400))) for _ in range(5)]
            sets = [oracle_sets(rs) for rs in row_sets]
            union = tuple(set().union(*[s[k] for s in sets]) for k in range(3))
            for flags in (0, 1):
                res = I.device_ingest(ctx, row_sets, FPR, parent_of_set=[0] * 5, n_parents=1, flags=flags)
                for s_, st in enumerate(sets):
                    check(res, s_, st, "seed %d generator %d flags %d set %d" % (seed, kind, flags, s_))
                check(res, 5, union, "seed %d generator %d flags %d file" % (seed, kind, flags))
                n_fb += len(res.fallback_rows)
            # the same rows damaged, through the validating walk only: whatever the device keeps of a row it cannot finish, the
            # result must be what the host walker alone produces (and nothing may hang)
            bad_sets = [mutate(rng, rs) for rs in row_sets]
            res = I.device_ingest(ctx, bad_sets, FPR, parent_of_set=[0] * 5, n_parents=1, flags=0)
            for s_, rs in enumerate(bad_sets):
                check(res, s_, host_sets(rs), "seed %d generator %d damaged rows set %d" % (seed, kind, s_))
            n_fb += len(res.fallback_rows)
            rows = [r for rs in row_sets for r in rs]
            n_rows += len(rows)
            vocab, paths = sorted(union[1]) or ["x"], sorted(union[0]) or ["x"]

            def rand_expr(depth=0):
                r = rng.random()
                if depth >= 3 or r < 0.5:
                    tok = vocab[rng.integers(0, len(vocab))] if rng.random() < 0.85 else "absent%d" % rng.integers(0, 99)
                    fld = paths[rng.integers(0, len(paths))] if rng.random() < 0.85 else "nope.%d" % rng.integers(0, 9)
                    return [Q.Field(fld), Q.Token(tok), Q.FieldToken(fld, tok)][rng.integers(0, 3)]
                kids = [rand_expr(depth + 1) for _ in range(int(rng.integers(0, 4)))]
                return Q.And(*kids) if rng.random() < 0.5 else Q.Or(*kids)
            for _ in range(6):
                e = rand_expr()
                got, fb = ctx.match_rows(rows, Q.CompiledMatcher(e))
                for r in fb:
                    got[r] = Hst.match_row(e, rows[int(r)])
                want = [Hst.match_row(e, r) for r in rows]
                if list(map(bool, got)) != want:
                    bad = [i for i, (a, b) in enumerate(zip(got, want)) if bool(a) != b][:3]
                    sys.exit("seed %d generator %d: k_match_rows differs from the host matcher on rows %s for %s" % (seed, kind, bad, json.dumps(e)))
                n_match += len(rows)
                if _ < 2:                                       # the damaged rows through the matcher too
                    bad_rows = [r for rs in bad_sets for r in rs]
                    got, fb = ctx.match_rows(bad_rows, Q.CompiledMatcher(e))
                    for r in fb:
                        got[r] = Hst.match_row(e, bad_rows[int(r)])
                    if list(map(bool, got)) != [Hst.match_row(e, r) for r in bad_rows]:
                        sys.exit("seed %d generator %d: k_match_rows differs from the host matcher on damaged rows for %s" % (seed, kind, json.dumps(e)))
                    n_match += len(bad_rows)
        print("seed %d ok (%d rows so far, %d handed to the host walker, %d row verdicts)" % (seed, n_rows, n_fb, n_match), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
