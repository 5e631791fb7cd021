"""Malformed rows, mutated rows, truncated rows, deep nesting and corrupted filter sections through the C++ host mirror
(walker, tokenizer, matchers, regex matcher, section codec).  Meant to run under AddressSanitizer / UBSan: tools/asan_host.sh."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bloomsearch_amd import _lib
if os.environ.get('BSG_LAB_LIB'):
    _lib.LIB_PATH = os.environ['BSG_LAB_LIB']
import numpy as np
from bloomsearch_amd import host as Hst, query as Q
from tests.test_host_tables import KEYS, _random_value, go_marshal

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
alphabet = list(b'{}[]":,\\ \t\n0123456789.-+eEtruefalsn') + [0, 0x7f, 0x80, 0xbf, 0xc2, 0xe0, 0xed, 0xf0, 0xf4, 0xff] + list(b'abcXYZ')
sets = Hst.EntrySets()
exprs = [Q.And(Q.Token("a"), Q.FieldToken("b.c", "x")), Q.Or(Q.Field("k"), Q.Token("true")), None]
rex = {"ExpressionType": "CONDITION", "Condition": {"Type": "FIELD_REGEX", "Field": "a", "Pattern": "^x+[0-9]*$"}}
n_ok = 0
for i in range(N):
    kind = rng.integers(0, 4)
    if kind == 0:      # random bytes from a JSON-ish alphabet
        row = bytes(alphabet[j] for j in rng.integers(0, len(alphabet), size=int(rng.integers(0, 80))))
    elif kind == 1:    # valid rows, then mutated
        row = bytearray(go_marshal({KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))}))
        for _ in range(int(rng.integers(0, 4))):
            if len(row):
                p = int(rng.integers(0, len(row)))
                op = rng.integers(0, 3)
                if op == 0: row[p] = int(rng.integers(0, 256))
                elif op == 1: del row[p]
                else: row.insert(p, int(rng.integers(0, 256)))
        row = bytes(row)
    elif kind == 2:    # truncated valid rows
        row = go_marshal({KEYS[rng.integers(0, len(KEYS))]: _random_value(rng, 0) for _ in range(rng.integers(1, 6))})
        row = row[: int(rng.integers(0, len(row) + 1))]
    else:              # deep nesting / long keys
        d = int(rng.integers(1, 60))
        row = (b'{"k":' * d) + b'1' + (b'}' * int(rng.integers(0, d + 2)))
    try:
        sets.index_row(row); n_ok += 1
    except Exception:
        pass
    for e in exprs:
        try: Hst.match_row(e, row)
        except Exception: pass
    try: Hst.match_row_regex(rex, row)
    except Exception: pass
    try: Hst.tokenize(row)
    except Exception: pass
# sections: valid then corrupted / truncated / random
for i in range(N // 4):
    fl = []
    for c in range(3):
        m = int(rng.integers(1, 2000)); k = int(rng.integers(1, 20))
        fl.append((m, k, rng.integers(0, 2**63, size=(m + 63) // 64, dtype=np.uint64)))
    sec = bytearray(Hst.section_encode(fl))
    mode = rng.integers(0, 4)
    if mode == 1 and len(sec): sec[int(rng.integers(0, len(sec)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 2: sec = sec[: int(rng.integers(0, len(sec) + 1))]
    elif mode == 3: sec = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8).tobytes())
    try: Hst.section_parse(bytes(sec))
    except Exception: pass
print("host fuzz done: %d rows (%d indexed), %d sections" % (N, n_ok, N // 4))
