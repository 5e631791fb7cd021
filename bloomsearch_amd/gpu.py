"""Thin numpy-facing wrapper over the C-ABI (include/bloomgpu.h).

Test / bench glue only: every method is one call into libbloomgpu.so; no
arithmetic of the hot path is done in Python.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import DESC_DTYPE, TERM_DTYPE, BloomGpuError, IngestStats, Timing


def pack_entries(entries):
    """list[bytes|str] -> (u8 blob, u32 offsets[n+1]) in the layout bsg_hash_entries / bsg_build take."""
    entries = [e.encode() if isinstance(e, str) else bytes(e) for e in entries]
    off = np.zeros(len(entries) + 1, dtype=np.uint32)
    if entries:
        off[1:] = np.cumsum([len(e) for e in entries], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(entries), dtype=np.uint8).copy() if entries else np.zeros(0, dtype=np.uint8)
    return blob, off


def survivor_list(row: np.ndarray, n_blocks: int) -> np.ndarray:
    """bsg_survivor_list: the surviving block indices of one query's survivor row, ascending (host arithmetic)."""
    L = _lib.load()
    row = np.ascontiguousarray(row, dtype=np.uint64)
    n = C.c_uint32()
    out = np.zeros(max(n_blocks, 1), dtype=np.uint32)
    rc = L.bsg_survivor_list(_lib._ptr(row), n_blocks, _lib._ptr(out), len(out), C.byref(n))
    if rc:
        raise BloomGpuError(rc, L.bsg_last_error(None).decode())
    return out[: n.value].copy()


def survivor_row_list(hdr: int, row: np.ndarray, n_blocks: int) -> np.ndarray:
    """bsg_survivor_row_list: the surviving block indices of one row of bsg_probe_many_rows, whatever its tag."""
    L = _lib.load()
    row = np.ascontiguousarray(row, dtype=np.uint64)
    n = C.c_uint32()
    out = np.zeros(max(n_blocks, 1), dtype=np.uint32)
    rc = L.bsg_survivor_row_list(int(hdr), _lib._ptr(row), n_blocks, _lib._ptr(out), len(out), C.byref(n))
    if rc:
        raise BloomGpuError(rc, L.bsg_last_error(None).decode())
    return out[: n.value].copy()


def rows_to_dense(hdr: np.ndarray, rows: np.ndarray, n_blocks: int, packed: bool = False) -> np.ndarray:
    """[n_queries] headers + [n_queries][G] row slots of bsg_probe_many_rows -> the dense [n_queries][G] bitsets of bsg_probe_many
    (numpy; tests and consumers that want the bitset form).  packed: the rows were written with BSG_PROBE_ROWS_PACKED (payloads of
    every run of 256 queries back to back from the run's first slot)."""
    G = (n_blocks + 63) // 64
    flat = np.asarray(rows, dtype=np.uint64).reshape(-1)
    out = np.zeros((len(hdr), G), dtype=np.uint64)
    full = np.full(G, ~np.uint64(0), dtype=np.uint64)
    if n_blocks & 63:
        full[-1] = np.uint64((1 << (n_blocks & 63)) - 1)
    if packed:      # byte headers: tag << 6 | a LIST row's count (hdr: the first len(hdr) BYTES of what was passed, see packed_headers)
        hdr = np.asarray(hdr)
        hb = hdr if hdr.dtype == np.uint8 else hdr.view(np.uint8)[: len(hdr)]          # (a u32 view of the buffer: its first len(hdr) bytes)
        tag, cnt = (hb >> 6).astype(np.uint32), (hb & 63).astype(np.uint32)
    else:
        tag, cnt = np.asarray(hdr, dtype=np.uint32) >> 30, np.asarray(hdr, dtype=np.uint32) & np.uint32(0x3FFFFFFF)
    out[tag == 1] = full
    size = np.where(tag == 2, (cnt.astype(np.int64) + 1) >> 1, np.where(tag == 3, G, 0))           # payload words per row
    if packed:
        start = np.zeros(len(hdr), dtype=np.int64)
        for r0 in range(0, len(hdr), 256):
            s = size[r0: r0 + 256]
            start[r0: r0 + 256] = r0 * G + np.cumsum(s) - s
    else:
        start = np.arange(len(hdr), dtype=np.int64) * G
    for q in np.nonzero(tag == 3)[0]:
        out[q] = flat[start[q]: start[q] + G]
    for q in np.nonzero(tag == 2)[0]:
        ids = flat[start[q]: start[q] + size[q]].view(np.uint32)[: int(cnt[q])].astype(np.int64)
        np.bitwise_or.at(out[q], ids >> 6, np.uint64(1) << (ids & 63).astype(np.uint64))
    return out


def packed_headers(hdr: np.ndarray, index: int, n_queries: int) -> np.ndarray:
    """The n_queries byte headers of arena (or device x arena) number `index` out of the header buffer of a BSG_PROBE_ROWS_PACKED call."""
    return hdr.view(np.uint8)[index * n_queries: (index + 1) * n_queries]


def estimate_parameters(n: int, p: float):
    """bloom/v3 EstimateParameters with New()'s clamps (host arithmetic, no GPU needed)."""
    L = _lib.load()
    m, k = C.c_uint64(), C.c_uint64()
    rc = L.bsg_estimate_parameters(n, p, C.byref(m), C.byref(k))
    if rc:
        raise BloomGpuError(rc, L.bsg_last_error(None).decode())
    return int(m.value), int(k.value)


class Context:
    """bsg_ctx: one or more gfx950 devices, per-device streams."""

    def __init__(self, device_ids=(0,)):
        self.L = _lib.load()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.L.bsg_open(ids, len(device_ids), C.byref(h))
        if rc:
            raise BloomGpuError(rc, self.L.bsg_last_error(None).decode())
        self.h = h
        self.n_devices = len(device_ids)
        self._dependents = weakref.WeakSet()     # objects that hold this handle (host.Engine): closed before the context is

    def _check(self, rc):
        if rc:
            raise BloomGpuError(rc, self.L.bsg_last_error(self.h).decode())

    def close(self):
        if self.h:
            for dep in list(self._dependents):
                dep.close()
            self.L.bsg_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(self.L.bsg_sync(self.h))

    def device_calls(self) -> np.ndarray:
        """Construct / match parts each device of the context has served so far."""
        out = np.zeros(self.n_devices, dtype=np.uint64)
        self._check(self.L.bsg_device_calls(self.h, _lib._ptr(out), len(out)))
        return out

    # ---- construct ----
    def hash_entries(self, blob: np.ndarray, off: np.ndarray) -> np.ndarray:
        n = len(off) - 1
        out = np.zeros((n, 4), dtype=np.uint64)
        self._check(self.L.bsg_hash_entries(self.h, _lib._ptr(blob), _lib._ptr(off), n, _lib._ptr(out)))
        return out

    def hash_strings(self, entries) -> np.ndarray:
        return self.hash_entries(*pack_entries(entries))

    def build(self, blob, off, filter_entry_start, desc, n_words: int) -> np.ndarray:
        assert desc.dtype == DESC_DTYPE
        out = np.empty(n_words, dtype=np.uint64)
        fs = np.ascontiguousarray(filter_entry_start, dtype=np.uint32)
        self._check(self.L.bsg_build(self.h, _lib._ptr(blob), _lib._ptr(off), len(off) - 1, _lib._ptr(fs),
                                     _lib._ptr(desc), len(desc), _lib._ptr(out), n_words))
        return out

    def build_hashed(self, h, filter_entry_start, desc, n_words: int) -> np.ndarray:
        assert desc.dtype == DESC_DTYPE
        h = np.ascontiguousarray(h, dtype=np.uint64)
        out = np.empty(n_words, dtype=np.uint64)
        fs = np.ascontiguousarray(filter_entry_start, dtype=np.uint32)
        self._check(self.L.bsg_build_hashed(self.h, _lib._ptr(h), len(h), _lib._ptr(fs), _lib._ptr(desc), len(desc),
                                            _lib._ptr(out), n_words))
        return out

    # ---- probe ----
    def arena_load(self, words: np.ndarray, desc: np.ndarray) -> int:
        assert desc.dtype == DESC_DTYPE and len(desc) % 3 == 0
        words = np.ascontiguousarray(words, dtype=np.uint64)
        aid = C.c_uint64()
        self._check(self.L.bsg_arena_load(self.h, _lib._ptr(words), len(words), _lib._ptr(desc), len(desc) // 3,
                                          C.byref(aid)))
        return int(aid.value)

    def arena_load_sections(self, sections):
        """sections: list[bytes] (b"" = block without a section) -> (arena_id, int32 status per block)."""
        off = np.zeros(len(sections) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in sections], dtype=np.uint64)
        region = np.frombuffer(b"".join(sections), dtype=np.uint8)
        status = np.zeros(max(len(sections), 1), dtype=np.int32)
        aid = C.c_uint64()
        self._check(self.L.bsg_arena_load_sections(self.h, _lib._ptr(region), len(region), _lib._ptr(off), len(sections),
                                                   _lib._ptr(status), C.byref(aid)))
        return int(aid.value), status[: len(sections)]

    def arena_stream_begin(self, sec_begin, sec_end) -> int:
        sb = np.ascontiguousarray(sec_begin, dtype=np.uint64)
        se = np.ascontiguousarray(sec_end, dtype=np.uint64)
        sid = C.c_uint64()
        self._check(self.L.bsg_arena_stream_begin(self.h, _lib._ptr(sb), _lib._ptr(se), len(sb), C.byref(sid)))
        return int(sid.value)

    def arena_stream_append(self, stream_id: int, file_offset: int, data: bytes):
        buf = np.frombuffer(data, dtype=np.uint8)
        self._check(self.L.bsg_arena_stream_append(self.h, stream_id, file_offset, _lib._ptr(buf), len(buf)))

    def arena_stream_finish(self, stream_id: int, n_blocks: int):
        status = np.zeros(max(n_blocks, 1), dtype=np.int32)
        aid = C.c_uint64()
        self._check(self.L.bsg_arena_stream_finish(self.h, stream_id, _lib._ptr(status), C.byref(aid)))
        return int(aid.value), status[:n_blocks]

    def arena_stream_abort(self, stream_id: int):
        self._check(self.L.bsg_arena_stream_abort(self.h, stream_id))

    def arena_free(self, arena_id: int):
        self._check(self.L.bsg_arena_free(self.h, arena_id))

    # ---- resident file arenas across queries (cache_api.inc) ----
    def set_arena_budget(self, n_bytes: int):
        self._check(self.L.bsg_set_arena_budget(self.h, int(n_bytes)))

    def file_arena_acquire(self, key: bytes, block_keys):
        """-> (lease, arena_id, rows) when the file's resident arena covers every block key, else (0, 0, None)."""
        bk = np.ascontiguousarray(block_keys, dtype=np.uint64)
        kb = np.frombuffer(key, dtype=np.uint8)
        rows = np.zeros(max(len(bk), 1), dtype=np.uint32)
        lease, aid = C.c_uint64(), C.c_uint64()
        self._check(self.L.bsg_file_arena_acquire(self.h, _lib._ptr(kb), len(kb), _lib._ptr(bk), len(bk), C.byref(lease), C.byref(aid), None, _lib._ptr(rows)))
        return (int(lease.value), int(aid.value), rows[: len(bk)]) if lease.value else (0, 0, None)

    def file_arena_have(self, key: bytes):
        """Block keys and section extents (file offsets) of the file's resident arena: three u64 arrays (empty when there is none)."""
        kb = np.frombuffer(key, dtype=np.uint8)
        n = C.c_uint32()
        self._check(self.L.bsg_file_arena_have(self.h, _lib._ptr(kb), len(kb), None, None, None, 0, C.byref(n)))
        while True:
            cap = int(n.value)
            keys, sb, se = (np.zeros(max(cap, 1), dtype=np.uint64) for _ in range(3))
            rc = self.L.bsg_file_arena_have(self.h, _lib._ptr(kb), len(kb), _lib._ptr(keys), _lib._ptr(sb), _lib._ptr(se), max(cap, 1), C.byref(n))
            if rc == 0 and int(n.value) <= max(cap, 1):
                m = int(n.value)
                return keys[:m], sb[:m], se[:m]
            if int(n.value) <= cap:             # a real failure, not a wider arena published in between
                self._check(rc)

    def file_arena_publish(self, key: bytes, arena_id: int, block_keys, sec_begin, sec_end, status=None):
        """-> (lease, resident): the cache owns the arena from here on."""
        kb = np.frombuffer(key, dtype=np.uint8)
        bk, sb, se = (np.ascontiguousarray(a, dtype=np.uint64) for a in (block_keys, sec_begin, sec_end))
        st = None if status is None else np.ascontiguousarray(status, dtype=np.int32)
        lease, res = C.c_uint64(), C.c_int32()
        self._check(self.L.bsg_file_arena_publish(self.h, _lib._ptr(kb), len(kb), arena_id, _lib._ptr(bk), _lib._ptr(sb), _lib._ptr(se), _lib._ptr(st), len(bk),
                                                  C.byref(lease), C.byref(res)))
        return int(lease.value), bool(res.value)

    def file_arena_release(self, lease: int):
        self._check(self.L.bsg_file_arena_release(self.h, lease))

    def file_arena_forget(self, key: bytes):
        kb = np.frombuffer(key, dtype=np.uint8)
        self._check(self.L.bsg_file_arena_forget(self.h, _lib._ptr(kb), len(kb)))

    def arena_cache_stats(self, reset: bool = False) -> dict:
        st = _lib.ArenaCacheStats()
        self._check(self.L.bsg_arena_cache_stats_read(self.h, C.byref(st), 1 if reset else 0))
        return {f: int(getattr(st, f)) for f, _ in st._fields_}

    def batch_create(self, terms: np.ndarray, prog_ops, prog_off) -> int:
        assert terms.dtype == TERM_DTYPE
        ops = np.ascontiguousarray(prog_ops, dtype=np.uint32)
        off = np.ascontiguousarray(prog_off, dtype=np.uint32)
        bid = C.c_uint64()
        self._check(self.L.bsg_batch_create(self.h, _lib._ptr(terms), len(terms), _lib._ptr(ops), _lib._ptr(off),
                                            len(off) - 1, C.byref(bid)))
        return int(bid.value)

    def batch_free(self, batch_id: int):
        self._check(self.L.bsg_batch_free(self.h, batch_id))

    def probe_batch(self, arena_id: int, batch_id: int, n_queries: int, n_blocks: int, flags: int = 0,
                    want_output: bool = True):
        out = np.zeros((n_queries, (n_blocks + 63) // 64), dtype=np.uint64) if want_output else None
        self._check(self.L.bsg_probe_batch(self.h, arena_id, batch_id, flags, _lib._ptr(out)))
        return out

    def probe_many(self, arena_ids, batch_id: int, flags: int = 0, n_queries: int = 0, n_blocks=None):
        """n_blocks (list, one per arena) given => synchronous, returns a list of survivor arrays."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        if n_blocks is None:
            self._check(self.L.bsg_probe_many(self.h, _lib._ptr(ids), len(ids), batch_id, flags, None))
            return None
        sizes = [n_queries * ((nb + 63) // 64) for nb in n_blocks]
        out = np.zeros(max(sum(sizes), 1), dtype=np.uint64)
        self._check(self.L.bsg_probe_many(self.h, _lib._ptr(ids), len(ids), batch_id, flags, _lib._ptr(out)))
        res, o = [], 0
        for sz, nb in zip(sizes, n_blocks):
            res.append(out[o: o + sz].reshape(n_queries, (nb + 63) // 64))
            o += sz
        return res

    def probe_many_dev(self, arena_ids, batch_id: int, d_out_ptr: int, flags: int = 0):
        """Survivors of every arena, back to back, at a device pointer (single-device contexts)."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        self._check(self.L.bsg_probe_many_dev(self.h, _lib._ptr(ids), len(ids), batch_id, flags, C.c_void_p(d_out_ptr)))

    def probe_many_into(self, arena_ids, batch_id: int, out: np.ndarray, flags: int = 0):
        """Synchronous probe of every arena; survivors written back to back into `out` (u64, e.g. pinned memory)."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        self._check(self.L.bsg_probe_many(self.h, _lib._ptr(ids), len(ids), batch_id, flags, C.c_void_p(out.ctypes.data)))

    def probe_many_rows(self, arena_ids, batch_id: int, rows: np.ndarray, hdr: np.ndarray, flags: int = 0):
        """bsg_probe_many_rows: a header per (arena, query) into `hdr` (u32) and, only where the tag needs it, block ids or words into
        the row's dense slot of `rows` (u64) — both PAGE-LOCKED arrays (pinned_array / host_register), written by the device."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        self._check(self.L.bsg_probe_many_rows(self.h, _lib._ptr(ids), len(ids), batch_id, flags, C.c_void_p(rows.ctypes.data),
                                               C.c_void_p(hdr.ctypes.data)))

    def survivor_rows_size(self, arena_ids, batch_id: int):
        """bsg_survivor_rows_size: (row words, headers) bsg_probe_many_rows writes for these arenas and this batch on this context."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        rw, hw = C.c_uint64(), C.c_uint64()
        self._check(self.L.bsg_survivor_rows_size(self.h, _lib._ptr(ids), len(ids), batch_id, C.byref(rw), C.byref(hw)))
        return int(rw.value), int(hw.value)

    def survivor_rows_list(self, arena_ids, batch_id: int, rows: np.ndarray, hdr: np.ndarray, arena_index: int, query: int, n_blocks: int,
                           packed: bool = False) -> np.ndarray:
        """bsg_survivor_rows_list: the ascending GLOBAL block indices of (arena, query) from the rows bsg_probe_many_rows left — on a
        context of several devices the shards' rows merged (global = local * n_devices + device)."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        out = np.zeros(max(n_blocks, 1), dtype=np.uint32)
        n = C.c_uint32()
        fn = self.L.bsg_survivor_rows_list_packed if packed else self.L.bsg_survivor_rows_list
        self._check(fn(self.h, _lib._ptr(ids), len(ids), batch_id, C.c_void_p(rows.ctypes.data), C.c_void_p(hdr.ctypes.data),
                       arena_index, query, C.c_void_p(out.ctypes.data), len(out), C.byref(n)))
        return out[: int(n.value)]

    def set_probe_group(self, max_arenas_per_launch: int):
        self._check(self.L.bsg_set_probe_group(self.h, max_arenas_per_launch))

    def set_ingest_chunk(self, n_bytes: int):
        self._check(self.L.bsg_set_ingest_chunk(self.h, n_bytes))

    def set_lab(self, key: int, value: int):
        self._check(self.L.bsg_set_lab(self.h, key, value))

    def set_spin_wait(self, microseconds: int):
        self._check(self.L.bsg_set_spin_wait(self.h, microseconds))

    def set_fuse_limit(self, max_arenas: int):
        self._check(self.L.bsg_set_fuse_limit(self.h, max_arenas))

    def set_gather_cost(self, bytes_per_probe: int):
        self._check(self.L.bsg_set_gather_cost(self.h, bytes_per_probe))

    def scope(self):
        """An error scope: an alias of this context with its own last-error slot (bsg_scope_open)."""
        h = C.c_void_p()
        self._check(self.L.bsg_scope_open(self.h, C.byref(h)))
        sc = Context.__new__(Context)
        sc.L, sc.h, sc.n_devices = self.L, h, self.n_devices
        sc._dependents = weakref.WeakSet()
        return sc

    def probe(self, arena_id: int, n_blocks: int, terms: np.ndarray, prog_ops, prog_off) -> np.ndarray:
        assert terms.dtype == TERM_DTYPE
        ops = np.ascontiguousarray(prog_ops, dtype=np.uint32)
        off = np.ascontiguousarray(prog_off, dtype=np.uint32)
        nq = len(off) - 1
        out = np.zeros((nq, (n_blocks + 63) // 64), dtype=np.uint64)
        self._check(self.L.bsg_probe(self.h, arena_id, _lib._ptr(terms), len(terms), _lib._ptr(ops), _lib._ptr(off),
                                     nq, _lib._ptr(out)))
        return out

    def query(self, arena_ids, n_blocks, cb, out: np.ndarray = None):
        """bsg_query: one call, strings in -> survivors out.  cb: query.CompiledBatch (term strings + kinds + programs);
        n_blocks: blocks per arena.  -> list of [n_queries, ceil(n_blocks_i / 64)] u64 arrays."""
        ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
        if not hasattr(cb, "_packed"):
            tb, to = pack_entries(cb.term_strings)
            ops, poff, kinds = cb.arrays()
            cb._packed = (tb, to, np.ascontiguousarray(kinds, dtype=np.uint32), ops, poff)
        tb, to, kinds, ops, poff = cb._packed
        nq = len(poff) - 1
        sizes = [nq * ((nb + 63) // 64) for nb in n_blocks]
        if out is None:
            out = np.zeros(max(sum(sizes), 1), dtype=np.uint64)
        self._check(self.L.bsg_query(self.h, _lib._ptr(ids), len(ids), _lib._ptr(tb), _lib._ptr(to), _lib._ptr(kinds), len(kinds),
                                     _lib._ptr(ops), _lib._ptr(poff), nq, _lib._ptr(out)))
        res, o = [], 0
        for sz, nb in zip(sizes, n_blocks):
            res.append(out[o: o + sz].reshape(nq, (nb + 63) // 64))
            o += sz
        return res

    def query_stats(self, reset: bool = True) -> dict:
        """How the eligible bsg_query calls were served since the last reset (the combiner of concurrent callers)."""
        t = _lib.QueryStats()
        self._check(self.L.bsg_query_stats_read(self.h, C.byref(t), 1 if reset else 0))
        return {f: int(getattr(t, f)) for f, _ in _lib.QueryStats._fields_}

    def timing_read(self, reset: bool = True) -> Timing:
        t = Timing()
        self._check(self.L.bsg_timing_read(self.h, C.byref(t), 1 if reset else 0))
        return t

    def set_timed_stride(self, stride: int):
        self._check(self.L.bsg_set_timed_stride(self.h, stride))

    def last_kernel_ms(self):
        b, h, dc = C.c_float(), C.c_float(), C.c_float()
        self._check(self.L.bsg_last_kernel_ms(self.h, C.byref(b), C.byref(h), C.byref(dc)))
        return float(b.value), float(h.value), float(dc.value)

    # ---- OR-reduce ----
    def or_reduce(self, arena_id: int, kind: int, n_words: int) -> np.ndarray:
        out = np.zeros(n_words, dtype=np.uint64)
        self._check(self.L.bsg_or_reduce(self.h, arena_id, kind, _lib._ptr(out), n_words))
        return out

    def or_reduce_dev(self, arena_id: int, kind: int, d_out_ptr: int, n_words: int):
        self._check(self.L.bsg_or_reduce_dev(self.h, arena_id, kind, C.c_void_p(d_out_ptr), n_words))

    # ---- RCCL OR all-reduce inside the library ----
    @staticmethod
    def comm_unique_id() -> bytes:
        L = _lib.load()
        buf = (C.c_uint8 * 128)()
        rc = L.bsg_comm_unique_id(buf)
        if rc:
            raise BloomGpuError(rc, L.bsg_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id=None, rank: int = 0, world: int = 0):
        """unique_id None: every device of this (multi-device) context becomes a rank; else join as `rank` of `world`."""
        if unique_id is None:
            self._check(self.L.bsg_comm_init(self.h, None, 0, 0))
        else:
            buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
            self._check(self.L.bsg_comm_init(self.h, buf, rank, world))

    def peer_access(self) -> np.ndarray:
        """[n][n] u8: 1 where device i reaches device j's memory directly (bsg_peer_access)."""
        n = self.n_devices
        m = np.zeros((n, n), dtype=np.uint8)
        self._check(self.L.bsg_peer_access(self.h, _lib._ptr(m), n))
        return m

    def comm_info(self):
        """(ranks, this rank, asked_the_library) as the communicator library itself reports them (ncclCommCount / ncclCommUserRank)."""
        w, r, lib = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self.L.bsg_comm_info(self.h, C.byref(w), C.byref(r), C.byref(lib)))
        return int(w.value), int(r.value), bool(lib.value)

    def comm_destroy(self):
        self._check(self.L.bsg_comm_destroy(self.h))

    def or_allreduce(self, arena_id: int, kind: int, n_words: int) -> np.ndarray:
        out = np.zeros(n_words, dtype=np.uint64)
        self._check(self.L.bsg_or_allreduce(self.h, arena_id, kind, _lib._ptr(out), n_words))
        return out

    def or_allreduce_dev(self, d_ptrs, n_words: int):
        arr = (C.c_void_p * len(d_ptrs))(*d_ptrs)
        self._check(self.L.bsg_or_allreduce_dev(self.h, arr, n_words))

    def last_or_ms(self) -> float:
        v = C.c_float()
        self._check(self.L.bsg_last_or_ms(self.h, C.byref(v)))
        return float(v.value)

    def or_words_dev(self, d_dst_ptr: int, d_src_ptr: int, n_words: int, n_src: int):
        self._check(self.L.bsg_or_words_dev(self.h, C.c_void_p(d_dst_ptr), C.c_void_p(d_src_ptr), n_words, n_src))

    # ---- device ingest (rows -> distinct entries -> counts -> bitsets) ----
    def ingest_rows(self, rows, set_first_row, parent_of_set=None, n_parents: int = 0, slots_hint=None, flags: int = 0) -> int:
        """rows: list[bytes] (or (u8 blob, u64 offsets[n+1])); returns the ingest id."""
        if isinstance(rows, tuple):
            blob, off = rows
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
            off = np.ascontiguousarray(off, dtype=np.uint64)
        else:
            off = np.zeros(len(rows) + 1, dtype=np.uint64)
            if rows:
                off[1:] = np.cumsum([len(r) for r in rows], dtype=np.uint64)
            blob = np.frombuffer(b"".join(rows), dtype=np.uint8)
        sfr = np.ascontiguousarray(set_first_row, dtype=np.uint32)
        pos = None if parent_of_set is None else np.ascontiguousarray(parent_of_set, dtype=np.uint32)
        hint = None if slots_hint is None else np.ascontiguousarray(slots_hint, dtype=np.uint32)
        out = C.c_uint64()
        self._check(self.L.bsg_ingest_rows(self.h, _lib._ptr(blob), _lib._ptr(off), len(off) - 1, _lib._ptr(sfr), len(sfr) - 1,
                                           _lib._ptr(pos), n_parents, _lib._ptr(hint), flags, C.byref(out)))
        return int(out.value)

    def ingest_fallback_rows(self, ingest_id: int) -> np.ndarray:
        n = C.c_uint32()
        self._check(self.L.bsg_ingest_fallback_rows(self.h, ingest_id, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            self._check(self.L.bsg_ingest_fallback_rows(self.h, ingest_id, _lib._ptr(out), n.value, C.byref(n)))
        return out

    def ingest_add_entries(self, ingest_id: int, entries, set_of_entry, kind_of_entry):
        blob, off = pack_entries(entries)
        so = np.ascontiguousarray(set_of_entry, dtype=np.uint32)
        ko = np.ascontiguousarray(kind_of_entry, dtype=np.uint32)
        self._check(self.L.bsg_ingest_add_entries(self.h, ingest_id, _lib._ptr(blob), _lib._ptr(off), len(off) - 1,
                                                  _lib._ptr(so), _lib._ptr(ko)))

    def ingest_finish(self, ingest_id: int, n_sets_total: int):
        """-> (counts u64 [n_sets_total, 3], status u32 [n_sets_total])"""
        counts = np.zeros((n_sets_total, 3), dtype=np.uint64)
        status = np.zeros(n_sets_total, dtype=np.uint32)
        self._check(self.L.bsg_ingest_finish(self.h, ingest_id, _lib._ptr(counts), _lib._ptr(status)))
        return counts, status

    def ingest_build(self, ingest_id: int, desc, n_words: int, out: np.ndarray = None) -> np.ndarray:
        """out: where the words go (u64, >= n_words; e.g. a view of pinned_array memory: the copy back is then plain DMA)."""
        desc = np.ascontiguousarray(desc, dtype=DESC_DTYPE)
        if out is None:
            out = np.empty(n_words, dtype=np.uint64)
        assert out.dtype == np.uint64 and len(out) >= n_words
        self._check(self.L.bsg_ingest_build(self.h, ingest_id, _lib._ptr(desc), _lib._ptr(out), n_words))
        return out

    def sections_size(self, desc) -> int:
        desc = np.ascontiguousarray(desc, dtype=DESC_DTYPE)
        total = C.c_uint64()
        self._check(self.L.bsg_sections_size(_lib._ptr(desc), len(desc) // 3, C.byref(total)))
        return int(total.value)

    def _split_sections(self, region: np.ndarray, sec_off: np.ndarray):
        raw = region.tobytes()
        return [raw[int(sec_off[i]): int(sec_off[i + 1])] for i in range(len(sec_off) - 1)]

    def build_sections(self, blob, off, filter_entry_start, desc, n_words: int):
        """bsg_build + encodeFilterSection on the device -> list of section bytes (one per block)."""
        desc = np.ascontiguousarray(desc, dtype=DESC_DTYPE)
        fstart = np.ascontiguousarray(filter_entry_start, dtype=np.uint32)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        region = np.zeros(self.sections_size(desc), dtype=np.uint8)
        sec_off = np.zeros(len(desc) // 3 + 1, dtype=np.uint64)
        self._check(self.L.bsg_build_sections(self.h, _lib._ptr(blob), _lib._ptr(off), len(off) - 1, _lib._ptr(fstart), _lib._ptr(desc),
                                              len(desc), n_words, _lib._ptr(region), len(region), _lib._ptr(sec_off)))
        return self._split_sections(region, sec_off)

    def ingest_build_sections(self, ingest_id: int, desc, arenas: bool = False):
        """-> sections (list of bytes, sets then parents); with arenas=True also (sets_arena_id, parents_arena_id)."""
        desc = np.ascontiguousarray(desc, dtype=DESC_DTYPE)
        region = np.zeros(self.sections_size(desc), dtype=np.uint8)
        sec_off = np.zeros(len(desc) // 3 + 1, dtype=np.uint64)
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.L.bsg_ingest_build_sections(self.h, ingest_id, _lib._ptr(desc), _lib._ptr(region), len(region),
                                                     _lib._ptr(sec_off), C.byref(a) if arenas else None, C.byref(b) if arenas else None))
        secs = self._split_sections(region, sec_off)
        return (secs, int(a.value), int(b.value)) if arenas else secs

    def last_encode_ms(self) -> float:
        v = C.c_float()
        self._check(self.L.bsg_last_encode_ms(self.h, C.byref(v)))
        return float(v.value)

    # ---- pinned host memory ----
    def pinned_array(self, n_bytes: int) -> np.ndarray:
        """A u8 numpy view of n_bytes of pinned host memory (freed by pinned_free(array))."""
        p = C.c_void_p()
        self._check(self.L.bsg_pinned_alloc(self.h, n_bytes, C.byref(p)))
        buf = (C.c_uint8 * max(n_bytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8, count=n_bytes)
        return arr

    def pinned_free(self, arr: np.ndarray):
        self._check(self.L.bsg_pinned_free(self.h, C.c_void_p(arr.ctypes.data)))

    def host_register(self, arr: np.ndarray):
        self._check(self.L.bsg_host_register(self.h, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def host_unregister(self, arr: np.ndarray):
        self._check(self.L.bsg_host_unregister(self.h, C.c_void_p(arr.ctypes.data)))

    # ---- final row test on the device ----
    def match_rows(self, rows, matcher):
        """rows: list[bytes] or (u8 blob, u64 offsets); matcher: query.CompiledMatcher.
        -> (bool array [n_rows], sorted u32 array of rows the host matcher must decide)."""
        if isinstance(rows, tuple):
            blob = np.ascontiguousarray(rows[0], dtype=np.uint8)
            off = np.ascontiguousarray(rows[1], dtype=np.uint64)
        else:
            off = np.zeros(len(rows) + 1, dtype=np.uint64)
            if rows:
                off[1:] = np.cumsum([len(r) for r in rows], dtype=np.uint64)
            blob = np.frombuffer(b"".join(rows), dtype=np.uint8)
        n = len(off) - 1
        strings = [s for pair in zip(matcher.fields, matcher.tokens) for s in pair]      # field 0, token 0, field 1, ...
        cblob, coff = pack_entries(strings)
        kinds = np.asarray(matcher.kinds, dtype=np.uint32)
        ops = np.asarray(matcher.prog_ops, dtype=np.uint32)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        fb = np.zeros(max(n, 1), dtype=np.uint32)
        nfb = C.c_uint32()
        self._check(self.L.bsg_match_rows(self.h, _lib._ptr(blob), _lib._ptr(off), n, _lib._ptr(cblob), _lib._ptr(coff), _lib._ptr(kinds), len(kinds),
                                          _lib._ptr(ops), len(ops), _lib._ptr(bits), _lib._ptr(fb), len(fb), C.byref(nfb)))
        match = np.unpackbits(bits.view(np.uint8), bitorder="little")[:n].astype(bool)
        return match, fb[: nfb.value].copy()

    def last_match_ms(self) -> float:
        v = C.c_float()
        self._check(self.L.bsg_last_match_ms(self.h, C.byref(v)))
        return float(v.value)

    def ingest_stats(self, ingest_id: int) -> IngestStats:
        st = IngestStats()
        self._check(self.L.bsg_ingest_stats_read(self.h, ingest_id, C.byref(st)))
        return st

    def ingest_free(self, ingest_id: int):
        self._check(self.L.bsg_ingest_free(self.h, ingest_id))
