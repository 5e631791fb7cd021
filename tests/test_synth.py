"""The synthetic workload's two faces agree: rows as marshaled JSON pushed through the host walker /
tokenizer (what IngestRows sees) give exactly the entry sets block_entry_sets() produces directly."""
import numpy as np

from bloomsearch_amd import host as Hst, synth
from oracle import walker_oracle as W


def _sets(parts):
    out = []
    for blob, ln in parts:
        off = np.concatenate([[0], np.cumsum(ln, dtype=np.int64)])
        raw = np.asarray(blob, dtype=np.uint8).tobytes()
        out.append({raw[off[i]: off[i + 1]].decode() for i in range(len(ln))})
    return tuple(out)


def test_rows_and_direct_entry_sets_agree():
    for r0, n in ((0, 300), (12345, 257)):
        rows = synth.rows_json(r0, n)
        s = Hst.EntrySets()
        sets = (set(), set(), set())
        for r in rows:
            s.index_row(r)
            W.index_row(r, sets)
        direct = _sets(synth.block_entry_sets(r0, n))
        assert s.as_python_sets() == direct
        assert sets == direct
        assert set(synth.FIELD_PATHS) == direct[0]


def test_draws_are_range_independent():
    a = synth.draws(100, 50)
    b = synth.draws(120, 10)
    for k in a:
        assert np.array_equal(a[k][20:30], b[k])
