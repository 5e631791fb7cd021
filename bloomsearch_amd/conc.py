"""Concurrent-caller harness: T native threads calling bsg_query on one context (tools/native/conc_driver.cpp).

Test / bench glue only.  The driver is a few lines of C++ over include/bloomgpu.h, built in-tree next to its source
(tools/native/libconc_driver.so, git-ignored like every built artefact) by __graft_entry__.build().
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

import numpy as np

from . import _lib
from .gpu import pack_entries

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "native", "conc_driver.cpp")
LIB = os.path.join(ROOT, "tools", "native", "libconc_driver.so")


class ConcQuery(C.Structure):
    _fields_ = [("term_bytes", C.c_void_p), ("term_off", C.c_void_p), ("term_kinds", C.c_void_p), ("n_terms", C.c_uint32),
                ("prog_ops", C.c_void_p), ("n_ops", C.c_uint32)]


class ConcResult(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("mismatches", C.c_uint64), ("errors", C.c_uint64), ("seconds", C.c_double), ("n_lat", C.c_uint64), ("cpu_seconds", C.c_double)]


class CacheFile(C.Structure):
    _fields_ = [("region", C.c_void_p), ("sec_off", C.c_void_p), ("n_blocks", C.c_uint32)]


class CacheResult(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("mismatches", C.c_uint64), ("errors", C.c_uint64), ("hits", C.c_uint64), ("misses", C.c_uint64),
                ("forgets", C.c_uint64), ("seconds", C.c_double)]


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cxx = shutil.which("g++") or shutil.which("hipcc")
    csrc = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-o", LIB + ".tmp", SRC,
                           "-L", csrc, "-lbloomgpu", "-Wl,-rpath," + csrc, "-lpthread"])
    os.replace(LIB + ".tmp", LIB)
    return LIB


_drv = None


def load():
    global _drv
    if _drv is None:
        _lib.load()                               # libbloomgpu first: the driver links against it
        if not os.path.exists(LIB):
            raise ImportError("%s is missing: __graft_entry__.build() compiles it" % LIB)
        _drv = C.CDLL(LIB)
        _drv.conc_run.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(ConcResult)]
        _drv.conc_run.restype = C.c_int32
        _drv.cache_run.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64,
                                   C.POINTER(CacheResult)]
        _drv.cache_run.restype = C.c_int32
    return _drv


def _marshal_queries(exprs):
    from . import query as Q
    keep, qs = [], (ConcQuery * len(exprs))()
    for i, e in enumerate(exprs):
        cb = Q.compile_queries([e])
        tb, to = pack_entries(cb.term_strings)
        ops, poff, kinds = cb.arrays()
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        kinds = np.ascontiguousarray(kinds, dtype=np.uint32)
        tb = np.ascontiguousarray(np.frombuffer(tb, dtype=np.uint8) if isinstance(tb, (bytes, bytearray)) else tb)
        to = np.ascontiguousarray(to, dtype=np.uint32)
        keep.append((tb, to, kinds, ops))
        qs[i] = ConcQuery(tb.ctypes.data if tb.size else None, to.ctypes.data, kinds.ctypes.data if kinds.size else None, len(kinds),
                          ops.ctypes.data if ops.size else None, len(ops))
    return keep, qs


def cache_run(ctx, exprs, files, expected, n_threads, seconds, forget_every=0, seed=1):
    """T native threads over the resident file-arena cache (tools/native/conc_driver.cpp::cache_run).  files: list of lists of section
    bytes (one list per file, one section per block); expected[f]: [len(exprs), ceil(blocks_f / 64)] survivors of the file's full block set."""
    keep, qs = _marshal_queries(exprs)
    cf, hold = (CacheFile * len(files))(), []
    for i, secs in enumerate(files):
        region = np.frombuffer(b"".join(secs), dtype=np.uint8)
        off = np.zeros(len(secs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in secs], dtype=np.uint64)
        hold.append((region, off))
        cf[i] = CacheFile(region.ctypes.data, off.ctypes.data, len(secs))
    exp = [np.ascontiguousarray(e, dtype=np.uint64) for e in expected]
    for e, secs in zip(exp, files):
        assert e.shape == (len(exprs), (len(secs) + 63) // 64), e.shape
    ptrs = (C.c_void_p * len(exp))(*[e.ctypes.data for e in exp])
    res = CacheResult()
    rc = load().cache_run(ctx.h, n_threads, float(seconds), C.cast(qs, C.c_void_p), len(exprs), C.cast(cf, C.c_void_p), len(files),
                          C.cast(ptrs, C.c_void_p), forget_every, seed, C.byref(res))
    assert rc == 0
    return {f: getattr(res, f) for f, _ in res._fields_}


def run(ctx, exprs, arena_ids, n_blocks, expected, n_threads, seconds, arenas_per_call=1, lat_cap=1 << 18):
    """T threads x bsg_query(one query of `exprs`, arenas_per_call arenas) for `seconds`.  expected: [len(exprs), G] survivors of
    every arena of the list.  -> dict(calls, queries_per_s, mismatches, errors, p50_us, p99_us)."""
    from . import query as Q
    keep, qs = [], (ConcQuery * len(exprs))()
    for i, e in enumerate(exprs):
        cb = Q.compile_queries([e])
        tb, to = pack_entries(cb.term_strings)
        ops, poff, kinds = cb.arrays()
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        kinds = np.ascontiguousarray(kinds, dtype=np.uint32)
        tb = np.ascontiguousarray(np.frombuffer(tb, dtype=np.uint8) if isinstance(tb, (bytes, bytearray)) else tb)
        to = np.ascontiguousarray(to, dtype=np.uint32)
        keep.append((tb, to, kinds, ops))
        qs[i] = ConcQuery(tb.ctypes.data if tb.size else None, to.ctypes.data, kinds.ctypes.data if kinds.size else None, len(kinds),
                          ops.ctypes.data if ops.size else None, len(ops))
    ids = np.ascontiguousarray(arena_ids, dtype=np.uint64)
    exp = np.ascontiguousarray(expected, dtype=np.uint64)
    assert exp.shape == (len(exprs), (n_blocks + 63) // 64)
    lat = np.zeros(lat_cap, dtype=np.uint64)
    res = ConcResult()
    rc = load().conc_run(ctx.h, n_threads, float(seconds), C.cast(qs, C.c_void_p), len(exprs), ids.ctypes.data, len(ids), arenas_per_call,
                         n_blocks, exp.ctypes.data, lat.ctypes.data, lat_cap, C.byref(res))
    assert rc == 0
    l = np.sort(lat[: int(res.n_lat)]) / 1e3
    return {"threads": n_threads, "arenas_per_call": arenas_per_call, "calls": int(res.calls), "seconds": float(res.seconds),
            "queries_per_s": res.calls / max(res.seconds, 1e-9), "mismatches": int(res.mismatches), "errors": int(res.errors),
            "cpu_us_per_call": res.cpu_seconds / max(res.calls, 1) * 1e6, "cpus_busy": res.cpu_seconds / max(res.seconds, 1e-9),
            "p50_us": float(l[len(l) // 2]) if len(l) else None, "p99_us": float(l[min(len(l) - 1, int(len(l) * 0.99))]) if len(l) else None}
