"""ctypes front-end of the CPU ORACLE (oracle/bloom_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from bloomsearch_amd/ (the product path).
Parity status: unpinned at the bit level by the reference (see the C header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbloom_oracle.so")

KIND_FIELD, KIND_TOKEN, KIND_FIELD_TOKEN = 0, 1, 2
OP_TERM, OP_AND, OP_OR, OP_TRUE, OP_FALSE = 0, 1, 2, 3, 4

TERM_DTYPE = np.dtype([("h", "<u8", (4,)), ("kind", "<u4"), ("reserved", "<u4")])
DESC_DTYPE = np.dtype([("word_off", "<u8"), ("m", "<u8"), ("k", "<u4"), ("reserved", "<u4")])


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "bloom_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.bo_murmur3_x64_128.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, u64p]
        L.bo_base_hashes.argtypes = [C.c_char_p, C.c_uint64, u64p]
        L.bo_location.argtypes = [u64p, C.c_uint64]
        L.bo_location.restype = C.c_uint64
        L.bo_estimate_parameters.argtypes = [C.c_uint64, C.c_double, u64p, u64p]
        L.bo_filter_add.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]
        L.bo_filter_test.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]
        L.bo_filter_test.restype = C.c_int
        L.bo_build.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.bo_filter_serialize.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.bo_filter_serialize.restype = C.c_uint64
        L.bo_filter_deserialize.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p, u64p, C.c_uint64]
        L.bo_filter_deserialize.restype = C.c_uint64
        L.bo_crc32c.argtypes = [C.c_void_p, C.c_uint64]
        L.bo_crc32c.restype = C.c_uint32
        L.bo_encode_filter_section.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_void_p), u64p, u64p, C.c_void_p]
        L.bo_encode_filter_section.restype = C.c_uint64
        L.bo_parse_filter_section.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_int), u64p, u64p, u64p]
        L.bo_parse_filter_section.restype = C.c_int
        L.bo_probe_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.bo_probe_reference_style.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                               C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.bo_probe_reference_style.restype = C.c_int
        L.bo_build_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.bo_test_string_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def _u64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def murmur3_x64_128(data: bytes, seed: int = 0):
    out = (C.c_uint64 * 2)()
    lib().bo_murmur3_x64_128(data, len(data), seed, out)
    return int(out[0]), int(out[1])


# ---- an INDEPENDENT MurmurHash3_x64_128: Austin Appleby's public-domain reference implementation ----
# scikit-learn ships the original MurmurHash3.cpp (sklearn/utils/src/); where that file exists it is compiled FROM WHERE IT
# LIES (nothing of it enters this repository) into oracle/_third_party/ and the restatement in bloom_oracle.c — and the HIP
# kernels — are compared with it on arbitrary inputs, not only on the three public vectors.  It pins the hash function of
# SURVEY 8c's assumption B2 for every length and tail; what it cannot pin is bloom/v3's use of it (the "data || 0x01" second
# hash, the location formula, the wire layout): those stay with the Go parity harness.
_APPLEBY = None


def appleby_source():
    try:
        import sklearn
    except Exception:  # noqa: BLE001 - absent package: the pin is skipped, nothing else depends on it
        return None
    src = os.path.join(os.path.dirname(sklearn.__file__), "utils", "src", "MurmurHash3.cpp")
    return src if os.path.exists(src) else None


def appleby():
    """ctypes handle on MurmurHash3_x64_128 (key, len, seed, out[2 x u64]) of the third-party reference, or None."""
    global _APPLEBY
    if _APPLEBY is not None:
        return _APPLEBY or None
    src = appleby_source()
    out_dir = os.path.join(_HERE, "_third_party")
    so = os.path.join(out_dir, "libmurmur3_appleby.so")
    if src is None and not os.path.exists(so):
        _APPLEBY = False
        return None
    if src is not None and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.dirname(src), "-o", so, src])
    L = C.CDLL(so)
    fn = None
    for name in ("MurmurHash3_x64_128", "_Z19MurmurHash3_x64_128PKvijPv"):      # (C or C++ linkage, whatever the header asks for)
        if hasattr(L, name):
            fn = getattr(L, name)
            break
    if fn is None:
        _APPLEBY = False
        return None
    fn.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_void_p]
    fn.restype = None
    _APPLEBY = fn
    return fn


def appleby_x64_128(data: bytes, seed: int = 0):
    out = (C.c_uint64 * 2)()
    appleby()(data, len(data), seed, out)
    return int(out[0]), int(out[1])


def base_hashes(data: bytes):
    out = (C.c_uint64 * 4)()
    lib().bo_base_hashes(data, len(data), out)
    return tuple(int(x) for x in out)


def location(h, i: int) -> int:
    arr = (C.c_uint64 * 4)(*h)
    return int(lib().bo_location(arr, i))


def estimate_parameters(n: int, p: float):
    m, k = C.c_uint64(), C.c_uint64()
    lib().bo_estimate_parameters(n, p, C.byref(m), C.byref(k))
    return int(m.value), int(k.value)


def words_for(m: int) -> int:
    return (m + 63) // 64


class Filter:
    """One bloom filter: (m, k, words LE-native u64) — mirrors bloom.BloomFilter."""

    def __init__(self, m: int, k: int, words: np.ndarray | None = None):
        self.m, self.k = int(m), int(k)
        self.words = np.zeros(words_for(m), dtype=np.uint64) if words is None else np.ascontiguousarray(words, dtype=np.uint64)

    @classmethod
    def with_estimates(cls, n: int, p: float) -> "Filter":
        return cls(*estimate_parameters(n, p))

    def add(self, s: bytes | str):
        b = s.encode() if isinstance(s, str) else s
        lib().bo_filter_add(_u64p(self.words), self.m, self.k, b, len(b))

    def test(self, s: bytes | str) -> bool:
        b = s.encode() if isinstance(s, str) else s
        return bool(lib().bo_filter_test(_u64p(self.words), self.m, self.k, b, len(b)))

    def serialize(self) -> bytes:
        out = np.zeros(24 + 8 * len(self.words), dtype=np.uint8)
        n = lib().bo_filter_serialize(_u64p(self.words), self.m, self.k, out.ctypes.data)
        return out[:n].tobytes()

    @classmethod
    def deserialize(cls, raw: bytes) -> "Filter":
        buf = np.frombuffer(raw, dtype=np.uint8)
        cap = max(1, len(raw) // 8)
        words = np.zeros(cap, dtype=np.uint64)
        m, k = C.c_uint64(), C.c_uint64()
        n = lib().bo_filter_deserialize(buf.ctypes.data, len(raw), C.byref(m), C.byref(k), _u64p(words), cap)
        if n == 0:
            raise ValueError("malformed filter bytes")
        return cls(m.value, k.value, words[: words_for(m.value)].copy())


def build_sized(entries, p: float) -> Filter:
    """buildSizedBloomFilter (ingest.go:139-145): n = max(len(set), 1)."""
    entries = [e.encode() if isinstance(e, str) else e for e in entries]
    f = Filter.with_estimates(max(len(entries), 1), p)
    for e in entries:
        f.add(e)
    return f


def pack_entries(entries):
    """list[bytes] -> (bytes u8 array, offsets u32[n+1])."""
    entries = [e.encode() if isinstance(e, str) else e for e in entries]
    off = np.zeros(len(entries) + 1, dtype=np.uint32)
    if entries:
        off[1:] = np.cumsum([len(e) for e in entries], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(entries), dtype=np.uint8).copy() if entries else np.zeros(0, dtype=np.uint8)
    return blob, off


def crc32c(data: bytes) -> int:
    buf = np.frombuffer(data, dtype=np.uint8)
    return int(lib().bo_crc32c(buf.ctypes.data if len(data) else None, len(data)))


_HWCRC = None


def hw_crc32c_fn():
    """CRC-32C by the CPU's SSE4.2 crc32 instruction (oracle/hw_crc32c.c), or None on a CPU without it: an implementation of
    the checksum independent of the oracle's table walk, of the host codec and of the library's kernels."""
    global _HWCRC
    if _HWCRC is not None:
        return _HWCRC or None
    try:
        if "sse4_2" not in open("/proc/cpuinfo").read():
            raise OSError("no sse4.2")
        out_dir = os.path.join(_HERE, "_third_party")
        so, src = os.path.join(out_dir, "libhwcrc32c.so"), os.path.join(_HERE, "hw_crc32c.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(out_dir, exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-msse4.2", "-shared", "-fPIC", "-o", so, src])
        fn = C.CDLL(so).hw_crc32c
        fn.argtypes = [C.c_void_p, C.c_size_t]
        fn.restype = C.c_uint32
        _HWCRC = fn
        return fn
    except Exception:  # noqa: BLE001 - the pin is skipped where the instruction or gcc is missing
        _HWCRC = False
        return None


def hw_crc32c(data) -> int:
    buf = np.frombuffer(data, dtype=np.uint8)
    return int(hw_crc32c_fn()(buf.ctypes.data if len(buf) else None, len(buf)))


def encode_filter_section(filters) -> bytes:
    """filters: [Filter|None] * 3 in field/token/fieldtoken order."""
    present = (C.c_int * 3)(*[1 if f is not None else 0 for f in filters])
    wp = (C.c_void_p * 3)(*[f.words.ctypes.data if f is not None else None for f in filters])
    m = (C.c_uint64 * 3)(*[f.m if f is not None else 0 for f in filters])
    k = (C.c_uint64 * 3)(*[f.k if f is not None else 0 for f in filters])
    size = 5 + sum(4 + 24 + 8 * len(f.words) for f in filters if f is not None)
    out = np.zeros(size, dtype=np.uint8)
    n = lib().bo_encode_filter_section(present, wp, m, k, out.ctypes.data)
    assert n == size
    return out.tobytes()


def parse_filter_section(section: bytes):
    """-> [Filter|None]*3; raises ValueError(code) like parseFilterSection's errors."""
    buf = np.frombuffer(section, dtype=np.uint8)
    present = (C.c_int * 3)()
    m, k, woff = (C.c_uint64 * 3)(), (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
    rc = lib().bo_parse_filter_section(buf.ctypes.data if len(section) else None, len(section), present, m, k, woff)
    if rc:
        raise ValueError(rc)
    out = []
    for c in range(3):
        if not present[c]:
            out.append(None)
            continue
        nw = words_for(m[c])
        words = np.frombuffer(section, dtype=">u8", count=nw, offset=woff[c]).astype(np.uint64)
        out.append(Filter(m[c], k[c], words))
    return out


def probe_batch(arena_words: np.ndarray, desc: np.ndarray, terms: np.ndarray,
                prog_ops: np.ndarray, prog_off: np.ndarray) -> np.ndarray:
    """survivors[q][ceil(n_blocks/64)] — evaluateBlockFilters' verdict for every (query, block)."""
    assert desc.dtype == DESC_DTYPE and terms.dtype == TERM_DTYPE
    n_blocks = len(desc) // 3
    nq = len(prog_off) - 1
    out = np.zeros((nq, (n_blocks + 63) // 64), dtype=np.uint64)
    arena_words = np.ascontiguousarray(arena_words, dtype=np.uint64)
    prog_ops = np.ascontiguousarray(prog_ops, dtype=np.uint32)
    prog_off = np.ascontiguousarray(prog_off, dtype=np.uint32)
    lib().bo_probe_batch(arena_words.ctypes.data, desc.ctypes.data, n_blocks,
                         terms.ctypes.data, len(terms), prog_ops.ctypes.data, prog_off.ctypes.data, nq,
                         out.ctypes.data)
    return out


def probe_reference_style(sections: bytes, sec_off: np.ndarray, max_words: int,
                          term_strings, term_kinds, prog_ops, prog_off, n_threads: int = 1) -> np.ndarray:
    n_blocks = len(sec_off) - 1
    nq = len(prog_off) - 1
    blob, toff = pack_entries(term_strings)
    kinds = np.ascontiguousarray(term_kinds, dtype=np.uint32)
    secs = np.frombuffer(sections, dtype=np.uint8)
    sec_off = np.ascontiguousarray(sec_off, dtype=np.uint64)
    prog_ops = np.ascontiguousarray(prog_ops, dtype=np.uint32)
    prog_off = np.ascontiguousarray(prog_off, dtype=np.uint32)
    out = np.zeros((nq, (n_blocks + 63) // 64), dtype=np.uint64)
    rc = lib().bo_probe_reference_style(secs.ctypes.data, sec_off.ctypes.data, n_blocks, max_words,
                                        blob.ctypes.data, toff.ctypes.data, kinds.ctypes.data,
                                        prog_ops.ctypes.data, prog_off.ctypes.data, nq, n_threads, out.ctypes.data)
    if rc:
        raise ValueError(rc)
    return out


def build_many(blob: np.ndarray, off: np.ndarray, filter_entry_start: np.ndarray, desc: np.ndarray,
               n_words: int, n_threads: int = 1) -> np.ndarray:
    words = np.zeros(n_words, dtype=np.uint64)
    lib().bo_build_many(blob.ctypes.data, off.ctypes.data, filter_entry_start.ctypes.data,
                        desc.ctypes.data, len(desc), n_threads, words.ctypes.data)
    return words


# ---- tree-walking evaluator: evaluateBloomFilters / evaluateBloomExpression / evaluateBloomCondition restated over the
# expression TREE (dicts in the reference's exported JSON shape), with no postfix program and nothing from the product's
# lowering (bloomsearch_amd.query.compile_queries) in between.  query_exec.go:75-159. ----
_COND_KIND = {"FIELD": KIND_FIELD, "TOKEN": KIND_TOKEN, "FIELD_TOKEN": KIND_FIELD_TOKEN}


def _probed_string(cond: dict):
    """(kind, string) TestString is called with for a known condition type, else None (query_exec.go:134-157)."""
    t = cond.get("Type")
    if t == "FIELD":
        return KIND_FIELD, cond.get("Field", "")
    if t == "TOKEN":
        return KIND_TOKEN, cond.get("Token", "")
    if t == "FIELD_TOKEN":
        return KIND_FIELD_TOKEN, cond.get("Field", "") + "::" + cond.get("Token", "")    # makeFieldTokenKey, tokenizer.go:509-511
    return None


def _enc(s):
    return s if isinstance(s, bytes) else s.encode("utf-8", "surrogatepass")


def evaluate_condition(filters, cond: dict) -> bool:
    """evaluateBloomCondition (query_exec.go:128-159): filters = [field, token, fieldToken] Filter|None."""
    ks = _probed_string(cond)
    if ks is None:
        return False                      # unknown condition type
    kind, s = ks
    f = filters[kind]
    if f is None:
        return True                       # nil filter cannot disqualify
    return f.test(_enc(s))


def evaluate_tree(filters, expr) -> bool:
    """evaluateBloomFilters + evaluateBloomExpression (query_exec.go:75-126), short-circuit order included."""
    if expr is None:
        return True
    et = expr.get("ExpressionType")
    if et == "CONDITION":
        cond = expr.get("Condition")
        return True if cond is None else evaluate_condition(filters, cond)
    if et == "OR":
        kids = expr.get("Children") or []
        if len(kids) == 0:
            return False
        for c in kids:
            if evaluate_tree(filters, c):
                return True
        return False
    if et == "AND":
        for c in expr.get("Children") or []:
            if not evaluate_tree(filters, c):
                return False
        return True
    return False


def block_filters(words: np.ndarray, desc: np.ndarray, b: int):
    """[Filter|None] * 3 of block b of an arena (views into words)."""
    out = []
    for c in range(3):
        d = desc[b * 3 + c]
        m = int(d["m"])
        out.append(None if m == 0 else Filter(m, int(d["k"]), words[int(d["word_off"]): int(d["word_off"]) + words_for(m)]))
    return out


def evaluate_tree_blocks(words: np.ndarray, desc: np.ndarray, expr, cache: dict | None = None) -> np.ndarray:
    """evaluate_tree for every block of an arena at once: bool[n_blocks].  The same recursion over the same tree; a leaf is
    one TestString per block (bo_test_string_blocks), And / Or combine the per-block booleans elementwise — which is what
    the short-circuit loops compute."""
    assert desc.dtype == DESC_DTYPE
    n_blocks = len(desc) // 3
    words = np.ascontiguousarray(words, dtype=np.uint64)
    if cache is None:
        cache = {}

    def leaf(kind, s):
        key = (kind, s)
        v = cache.get(key)
        if v is None:
            raw = _enc(s)
            out = np.zeros(max(n_blocks, 1), dtype=np.uint8)
            lib().bo_test_string_blocks(words.ctypes.data, desc.ctypes.data, n_blocks, kind, raw, len(raw), out.ctypes.data)
            v = cache[key] = out[:n_blocks].astype(bool)
        return v

    def walk(e):
        if e is None:
            return np.ones(n_blocks, dtype=bool)
        et = e.get("ExpressionType")
        if et == "CONDITION":
            cond = e.get("Condition")
            if cond is None:
                return np.ones(n_blocks, dtype=bool)
            ks = _probed_string(cond)
            return np.zeros(n_blocks, dtype=bool) if ks is None else leaf(*ks)
        if et == "OR":
            acc = np.zeros(n_blocks, dtype=bool)
            for c in e.get("Children") or []:
                acc = acc | walk(c)
            return acc
        if et == "AND":
            acc = np.ones(n_blocks, dtype=bool)
            for c in e.get("Children") or []:
                acc = acc & walk(c)
            return acc
        return np.zeros(n_blocks, dtype=bool)

    return walk(expr)


def survivors_tree(words: np.ndarray, desc: np.ndarray, exprs) -> np.ndarray:
    """Survivor bitsets [len(exprs)][ceil(n_blocks / 64)] (the layout bsg_probe returns) from the tree-walking evaluator."""
    n_blocks = len(desc) // 3
    G = (n_blocks + 63) // 64
    out = np.zeros((len(exprs), G), dtype=np.uint64)
    cache: dict = {}
    for q, e in enumerate(exprs):
        v = evaluate_tree_blocks(words, desc, e, cache)
        bits = np.zeros(G * 64, dtype=np.uint8)
        bits[:n_blocks] = v
        out[q] = np.packbits(bits, bitorder="little").view(np.uint64)
    return out
