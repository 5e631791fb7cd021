"""ctypes wrapper over include/bloomsearch_host.h — the C++ host-side mirror of the reference's
tokenizer / walker / entry sets / query algebra / section codec / engine surface.
Test and bench glue only."""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import TERM_DTYPE

HOST_EXPORTS = [
    "bsh_free", "bsh_tokenize", "bsh_entry_sets_new", "bsh_entry_sets_free", "bsh_entry_sets_index_row",
    "bsh_entry_sets_union_into", "bsh_entry_sets_counts", "bsh_entry_sets_export_sizes", "bsh_entry_sets_export",
    "bsh_batch_new", "bsh_batch_free", "bsh_batch_add_query", "bsh_batch_sizes", "bsh_batch_export", "bsh_match_row", "bsh_prune_query", "bsh_match_row_regex",
    "bsh_section_encode", "bsh_section_parse", "bsh_crc32c",
    "bse_open", "bse_close", "bse_last_error", "bse_stop", "bse_ingest_rows", "bse_flush", "bse_merge", "bse_query",
    "bse_describe", "bse_corrupt_section_byte", "bse_section_bytes",
]

_H = None


class HostError(RuntimeError):
    def __init__(self, code, message=""):
        super().__init__(f"host error {code}: {message}")
        self.code = code


def lib():
    global _H
    if _H is not None:
        return _H
    L = _lib.load()
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    pp, pu64 = C.POINTER(vp), C.POINTER(u64)
    L.bsh_free.argtypes = [vp]; L.bsh_free.restype = None
    L.bsh_tokenize.argtypes = [C.c_char_p, u64, pp, pu64]
    L.bsh_entry_sets_new.restype = vp
    L.bsh_entry_sets_free.argtypes = [vp]; L.bsh_entry_sets_free.restype = None
    L.bsh_entry_sets_index_row.argtypes = [vp, C.c_char_p, u64]
    L.bsh_entry_sets_union_into.argtypes = [vp, vp]
    L.bsh_entry_sets_counts.argtypes = [vp, pu64]; L.bsh_entry_sets_counts.restype = None
    L.bsh_entry_sets_export_sizes.argtypes = [vp, u32, pu64, pu64]
    L.bsh_entry_sets_export.argtypes = [vp, u32, vp, vp]
    L.bsh_batch_new.restype = vp
    L.bsh_batch_free.argtypes = [vp]; L.bsh_batch_free.restype = None
    L.bsh_batch_add_query.argtypes = [vp, C.c_char_p, u64]
    L.bsh_batch_sizes.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), pu64]; L.bsh_batch_sizes.restype = None
    L.bsh_batch_export.argtypes = [vp, vp, vp, vp, vp, vp]
    L.bsh_match_row.argtypes = [C.c_char_p, u64, C.c_char_p, u64]
    L.bsh_prune_query.argtypes = [C.c_char_p, u64, C.c_char_p, u64, pp, pu64]
    L.bsh_match_row_regex.argtypes = [C.c_char_p, u64, C.c_char_p, u64]
    L.bsh_section_encode.argtypes = [C.POINTER(vp), pu64, pu64, pp, pu64]
    L.bsh_section_parse.argtypes = [C.c_char_p, u64, pu64, pu64, C.POINTER(vp)]
    L.bsh_crc32c.argtypes = [C.c_char_p, u64]; L.bsh_crc32c.restype = u32
    L.bse_open.argtypes = [C.c_char_p, u64, vp, pp]
    L.bse_close.argtypes = [vp]; L.bse_close.restype = None
    L.bse_last_error.argtypes = [vp]; L.bse_last_error.restype = C.c_char_p
    L.bse_stop.argtypes = [vp]
    L.bse_ingest_rows.argtypes = [vp, C.c_char_p, u64]
    L.bse_flush.argtypes = [vp]
    L.bse_merge.argtypes = [vp]
    L.bse_query.argtypes = [vp, C.c_char_p, u64, pp, pu64]
    L.bse_describe.argtypes = [vp, pp, pu64]
    L.bse_corrupt_section_byte.argtypes = [vp, u32, i32, u64]
    L.bse_section_bytes.argtypes = [vp, u32, i32, pp, pu64]
    for n in HOST_EXPORTS:
        f = getattr(L, n)
        if f.restype is C.c_int:
            f.restype = i32
    _H = L
    return L


def _take(L, p, n):
    data = C.string_at(p, n.value)
    L.bsh_free(p)
    return data


def tokenize(text: bytes | str) -> list[str]:
    L = lib()
    b = text.encode("utf-8", "surrogatepass") if isinstance(text, str) else text
    p, n = C.c_void_p(), C.c_uint64()
    rc = L.bsh_tokenize(b, len(b), C.byref(p), C.byref(n))
    if rc:
        raise HostError(rc)
    raw = _take(L, p, n)
    return [t.decode("utf-8", "replace") for t in raw.split(b"\n")] if raw else []


class EntrySets:
    """bloomEntrySets (ingest.go:24-123)."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.bsh_entry_sets_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.bsh_entry_sets_free(self.h)
            self.h = None

    def index_row(self, row: bytes):
        rc = self.L.bsh_entry_sets_index_row(self.h, row, len(row))
        if rc:
            raise HostError(rc, "row is not valid JSON")

    def union_into(self, dst: "EntrySets"):
        self.L.bsh_entry_sets_union_into(self.h, dst.h)

    def counts(self):
        c = (C.c_uint64 * 3)()
        self.L.bsh_entry_sets_counts(self.h, c)
        return tuple(int(x) for x in c)

    def export(self, kind: int):
        """-> (u8 blob, u32 lengths) in the layout arena.plan_blocks takes."""
        n, nb = C.c_uint64(), C.c_uint64()
        self.L.bsh_entry_sets_export_sizes(self.h, kind, C.byref(n), C.byref(nb))
        blob = np.zeros(nb.value, dtype=np.uint8)
        off = np.zeros(n.value + 1, dtype=np.uint32)
        self.L.bsh_entry_sets_export(self.h, kind, _lib._ptr(blob) or 0, off.ctypes.data)
        return blob, np.diff(off).astype(np.uint32)

    def as_python_sets(self):
        out = []
        for kind in range(3):
            blob, ln = self.export(kind)
            off = np.concatenate([[0], np.cumsum(ln, dtype=np.int64)]).astype(np.int64)
            raw = blob.tobytes()
            out.append({raw[off[i]: off[i + 1]].decode("utf-8", "surrogatepass") for i in range(len(ln))})
        return tuple(out)


class HostBatch:
    """QueryBatch: expression trees (reference JSON shape) lowered to C-ABI terms + programs."""

    def __init__(self, expressions=()):
        self.L = lib()
        self.h = C.c_void_p(self.L.bsh_batch_new())
        for e in expressions:
            self.add_query(e)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.bsh_batch_free(self.h)
            self.h = None

    def add_query(self, expression):
        s = json.dumps(expression).encode()
        rc = self.L.bsh_batch_add_query(self.h, s, len(s))
        if rc:
            raise HostError(rc, "bad expression JSON")

    def export(self):
        nq, nt, no, tb = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        self.L.bsh_batch_sizes(self.h, C.byref(nq), C.byref(nt), C.byref(no), C.byref(tb))
        blob = np.zeros(max(tb.value, 1), dtype=np.uint8)
        toff = np.zeros(nt.value + 1, dtype=np.uint32)
        kinds = np.zeros(max(nt.value, 1), dtype=np.uint32)
        ops = np.zeros(max(no.value, 1), dtype=np.uint32)
        poff = np.zeros(nq.value + 1, dtype=np.uint32)
        self.L.bsh_batch_export(self.h, blob.ctypes.data, toff.ctypes.data, kinds.ctypes.data, ops.ctypes.data, poff.ctypes.data)
        raw = blob.tobytes()
        strings = [raw[toff[i]: toff[i + 1]] for i in range(nt.value)]
        return strings, kinds[: nt.value], ops[: no.value], poff


def match_row(expression, row: bytes) -> bool:
    L = lib()
    s = json.dumps(expression).encode()
    rc = L.bsh_match_row(s, len(s), row, len(row))
    if rc < 0:
        raise HostError(rc)
    return bool(rc)


def prune_query(bloom_expression, regex_expression):
    """pruneBloomQuery of the host mirror -> expression dict or None."""
    L = lib()
    b = json.dumps(bloom_expression).encode()
    r = json.dumps(regex_expression).encode()
    p, n = C.c_void_p(), C.c_uint64()
    rc = L.bsh_prune_query(b, len(b), r, len(r), C.byref(p), C.byref(n))
    if rc:
        raise HostError(rc)
    return json.loads(_take(L, p, n))


def match_row_regex(regex_expression, row: bytes) -> bool:
    L = lib()
    r = json.dumps(regex_expression).encode()
    rc = L.bsh_match_row_regex(r, len(r), row, len(row))
    if rc < 0:
        raise HostError(rc)
    return bool(rc)


def crc32c(data: bytes) -> int:
    return int(lib().bsh_crc32c(data, len(data)))


def section_encode(filters) -> bytes:
    """filters: 3 x (m, k, words ndarray) or None."""
    L = lib()
    words = (C.c_void_p * 3)(*[f[2].ctypes.data if f is not None else None for f in filters])
    m = (C.c_uint64 * 3)(*[f[0] if f is not None else 0 for f in filters])
    k = (C.c_uint64 * 3)(*[f[1] if f is not None else 0 for f in filters])
    p, n = C.c_void_p(), C.c_uint64()
    rc = L.bsh_section_encode(words, m, k, C.byref(p), C.byref(n))
    if rc:
        raise HostError(rc)
    return _take(L, p, n)


def section_parse(section: bytes):
    L = lib()
    m, k = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
    w = (C.c_void_p * 3)()
    rc = L.bsh_section_parse(section, len(section), m, k, w)
    if rc:
        raise HostError(rc, "parseFilterSection failed")
    out = []
    for c in range(3):
        if not w[c]:
            out.append(None)
            continue
        nw = (m[c] + 63) // 64
        arr = np.frombuffer(C.string_at(w[c], nw * 8), dtype=np.uint64).copy()
        L.bsh_free(w[c])
        out.append((int(m[c]), int(k[c]), arr))
    return out


class Engine:
    """BloomSearchEngine mirror (IngestRows / Flush / Query / Merge) over a gpu.Context."""

    def __init__(self, ctx, **config):
        self.L = lib()
        self.ctx = ctx
        cfg = json.dumps(config).encode()
        h = C.c_void_p()
        rc = self.L.bse_open(cfg, len(cfg), ctx.h, C.byref(h))
        if rc:
            raise HostError(rc, "ErrInvalidConfig" if rc == -101 else "")
        self.h = h
        ctx._dependents.add(self)            # an engine that outlives its context (a traceback keeps it) must not call into freed memory

    def close(self):
        if getattr(self, "h", None):
            self.L.bse_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc:
            raise HostError(rc, self.L.bse_last_error(self.h).decode())

    def ingest_rows(self, rows):
        """rows: iterable of marshaled-JSON bytes (one object each)."""
        blob = b"\n".join(rows)
        self._check(self.L.bse_ingest_rows(self.h, blob, len(blob)))

    def flush(self):
        self._check(self.L.bse_flush(self.h))

    def merge(self):
        self._check(self.L.bse_merge(self.h))

    def stop(self):
        self._check(self.L.bse_stop(self.h))

    def query(self, bloom_expression=None, regex_expression=None):
        q = json.dumps({"Bloom": {"Expression": bloom_expression} if bloom_expression is not None else None,
                        "Regex": {"Expression": regex_expression} if regex_expression is not None else None}).encode()
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.bse_query(self.h, q, len(q), C.byref(p), C.byref(n)))
        return json.loads(_take(self.L, p, n))

    def describe(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.bse_describe(self.h, C.byref(p), C.byref(n)))
        return json.loads(_take(self.L, p, n))

    def corrupt_section_byte(self, file_index: int, block_index: int, byte_index: int):
        self._check(self.L.bse_corrupt_section_byte(self.h, file_index, block_index, byte_index))

    def section_bytes(self, file_index: int, block_index: int = -1) -> bytes:
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.bse_section_bytes(self.h, file_index, block_index, C.byref(p), C.byref(n)))
        return _take(self.L, p, n)
