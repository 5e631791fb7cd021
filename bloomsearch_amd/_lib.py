"""ctypes binding of libbloomgpu.so (include/bloomgpu.h).

Fails loudly: if the shared library is missing, or a compute entry point is
called without a gfx950 GPU, an exception is raised — there is no CPU fallback
in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libbloomgpu.so")

# status codes (bloomgpu.h)
BSG_OK, BSG_E_INVALID, BSG_E_HIP, BSG_E_NOMEM, BSG_E_NOTFOUND, BSG_E_UNSUPPORTED, BSG_E_NODEVICE = 0, -1, -2, -3, -4, -5, -6
KIND_FIELD, KIND_TOKEN, KIND_FIELD_TOKEN = 0, 1, 2
OP_TERM, OP_AND, OP_OR, OP_TRUE, OP_FALSE = 0, 1, 2, 3, 4
PROBE_ASYNC, PROBE_TIMED, PROBE_NOFUSE, PROBE_ROWS_PACKED = 1, 2, 4, 8
INGEST_TRUSTED_JSON = 1

TERM_DTYPE = np.dtype([("h", "<u8", (4,)), ("kind", "<u4"), ("reserved", "<u4")])
DESC_DTYPE = np.dtype([("word_off", "<u8"), ("m", "<u8"), ("k", "<u4"), ("reserved", "<u4")])


class Timing(C.Structure):
    _fields_ = [("n_probes", C.c_uint64), ("ms_terms_kernel", C.c_double), ("ms_eval_kernel", C.c_double),
                ("stream_bytes", C.c_uint64), ("n_probe_arenas", C.c_uint64), ("n_eval", C.c_uint64),
                ("n_fused", C.c_uint64), ("ms_fused_kernel", C.c_double), ("fused_stream_bytes", C.c_uint64),
                ("n_fused_arenas", C.c_uint64),
                ("n_folded", C.c_uint64), ("ms_folded_kernel", C.c_double), ("folded_stream_bytes", C.c_uint64),
                ("n_folded_arenas", C.c_uint64)]


class QueryStats(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("solo_calls", C.c_uint64), ("cycles", C.c_uint64), ("cycle_calls", C.c_uint64),
                ("dispatches", C.c_uint64), ("hot_arenas", C.c_uint64), ("max_calls_per_cycle", C.c_uint64),
                ("ns_prepare", C.c_uint64), ("ns_enqueue", C.c_uint64), ("ns_wait", C.c_uint64), ("ns_deal", C.c_uint64), ("ns_wake", C.c_uint64),
                ("ns_scatter", C.c_uint64), ("ns_free", C.c_uint64), ("ns_retire", C.c_uint64)]


class ArenaCacheStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("budget_bytes", "resident_bytes", "resident_files", "leases", "leased_dead_bytes", "hits", "misses", "published",
                                          "widenings", "evictions", "forgotten", "rejected_dirty", "rejected_over_budget", "rejected_narrower")]


class IngestStats(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("n_fallback_rows", C.c_uint32), ("table_grows", C.c_uint32), ("reserved", C.c_uint32),
                ("row_bytes", C.c_uint64), ("table_bytes", C.c_uint64), ("ms_walk", C.c_float), ("ms_union", C.c_float),
                ("ms_build", C.c_float), ("ms_encode", C.c_float)]


class BloomGpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libbloomgpu error {code}: {message}")
        self.code = code


def op(opcode: int, arg: int = 0) -> int:
    return (opcode << 28) | (arg & 0x0FFFFFFF)


# every symbol include/bloomgpu_lab.h declares: lab switches, not part of the drop-in contract
LAB_EXPORTS = ["bsg_set_lab", "bsg_set_spin_wait", "bsg_set_fuse_limit", "bsg_set_gather_cost", "bsg_lab_query_cpu", "bsg_set_timed_stride"]

# every symbol include/bloomgpu.h declares (tests assert the .so exports all of them)
EXPORTS = [
    "bsg_device_count", "bsg_peer_access", "bsg_device_calls", "bsg_open", "bsg_open_err", "bsg_close", "bsg_last_error", "bsg_last_error_copy", "bsg_scope_open",
    "bsg_sync", "bsg_estimate_parameters", "bsg_probe_many_dev", "bsg_set_probe_group", "bsg_set_ingest_chunk",
    "bsg_hash_entries", "bsg_build", "bsg_build_hashed", "bsg_arena_load", "bsg_arena_load_sections", "bsg_arena_free",
    "bsg_arena_stream_begin", "bsg_arena_stream_append", "bsg_arena_stream_finish", "bsg_arena_stream_abort",
    "bsg_set_arena_budget", "bsg_file_arena_acquire", "bsg_file_arena_have", "bsg_file_arena_publish", "bsg_file_arena_release",
    "bsg_file_arena_forget", "bsg_arena_cache_stats_read",
    "bsg_batch_create", "bsg_batch_free", "bsg_probe_batch", "bsg_probe_many", "bsg_probe", "bsg_query", "bsg_query_stats_read", "bsg_survivor_list", "bsg_probe_many_rows", "bsg_survivor_row_list", "bsg_survivor_rows_size", "bsg_survivor_rows_list", "bsg_survivor_rows_list_packed", "bsg_timing_read", "bsg_last_kernel_ms",
    "bsg_or_reduce", "bsg_or_words_dev", "bsg_or_reduce_dev", "bsg_last_or_ms",
    "bsg_comm_unique_id", "bsg_comm_init", "bsg_comm_destroy", "bsg_comm_info", "bsg_or_allreduce", "bsg_or_allreduce_dev",
    "bsg_ingest_rows", "bsg_ingest_fallback_rows", "bsg_ingest_add_entries", "bsg_ingest_finish", "bsg_ingest_build",
    "bsg_ingest_stats_read", "bsg_ingest_free", "bsg_ingest_build_sections",
    "bsg_sections_size", "bsg_build_sections", "bsg_last_encode_ms",
    "bsg_match_rows", "bsg_last_match_ms", "bsg_pinned_alloc", "bsg_pinned_free", "bsg_host_register", "bsg_host_unregister",
]

_lib = None


def load():
    """Load libbloomgpu.so; raises ImportError if it has not been built (python -m bloomsearch_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m bloomsearch_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.bsg_device_count.restype = i32
    L.bsg_open.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    L.bsg_device_calls.argtypes = [vp, vp, u32]
    L.bsg_peer_access.argtypes = [vp, vp, u32]
    L.bsg_open_err.argtypes = [C.POINTER(i32), i32, C.POINTER(vp), C.c_char_p, u64]
    L.bsg_last_error_copy.argtypes = [vp, C.c_char_p, u64]
    L.bsg_scope_open.argtypes = [vp, C.POINTER(vp)]
    L.bsg_probe_many_dev.argtypes = [vp, vp, u32, u64, u32, vp]
    L.bsg_set_probe_group.argtypes = [vp, u32]
    L.bsg_set_gather_cost.argtypes = [vp, u32]
    L.bsg_set_fuse_limit.argtypes = [vp, u32]
    L.bsg_set_spin_wait.argtypes = [vp, u32]
    L.bsg_set_ingest_chunk.argtypes = [vp, u64]
    L.bsg_set_lab.argtypes = [vp, u32, u64]
    L.bsg_close.argtypes = [vp]
    L.bsg_last_error.argtypes = [vp]
    L.bsg_last_error.restype = C.c_char_p
    L.bsg_sync.argtypes = [vp]
    L.bsg_estimate_parameters.argtypes = [u64, C.c_double, C.POINTER(u64), C.POINTER(u64)]
    L.bsg_hash_entries.argtypes = [vp, vp, vp, u32, vp]
    L.bsg_build.argtypes = [vp, vp, vp, u32, vp, vp, u32, vp, u64]
    L.bsg_build_hashed.argtypes = [vp, vp, u32, vp, vp, u32, vp, u64]
    L.bsg_arena_load.argtypes = [vp, vp, u64, vp, u32, C.POINTER(u64)]
    L.bsg_arena_load_sections.argtypes = [vp, vp, u64, vp, u32, vp, C.POINTER(u64)]
    L.bsg_arena_free.argtypes = [vp, u64]
    L.bsg_arena_stream_begin.argtypes = [vp, vp, vp, u32, C.POINTER(u64)]
    L.bsg_arena_stream_append.argtypes = [vp, u64, u64, vp, u64]
    L.bsg_arena_stream_finish.argtypes = [vp, u64, vp, C.POINTER(u64)]
    L.bsg_arena_stream_abort.argtypes = [vp, u64]
    L.bsg_set_arena_budget.argtypes = [vp, u64]
    L.bsg_file_arena_acquire.argtypes = [vp, vp, u32, vp, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), vp]
    L.bsg_file_arena_have.argtypes = [vp, vp, u32, vp, vp, vp, u32, C.POINTER(u32)]
    L.bsg_file_arena_publish.argtypes = [vp, vp, u32, u64, vp, vp, vp, vp, u32, C.POINTER(u64), C.POINTER(i32)]
    L.bsg_file_arena_release.argtypes = [vp, u64]
    L.bsg_file_arena_forget.argtypes = [vp, vp, u32]
    L.bsg_arena_cache_stats_read.argtypes = [vp, C.POINTER(ArenaCacheStats), i32]
    L.bsg_batch_create.argtypes = [vp, vp, u32, vp, vp, u32, C.POINTER(u64)]
    L.bsg_batch_free.argtypes = [vp, u64]
    L.bsg_probe_batch.argtypes = [vp, u64, u64, u32, vp]
    L.bsg_probe_many.argtypes = [vp, vp, u32, u64, u32, vp]
    L.bsg_probe.argtypes = [vp, u64, vp, u32, vp, vp, u32, vp]
    L.bsg_probe_many_rows.argtypes = [vp, vp, u32, u64, u32, vp, vp]
    L.bsg_survivor_row_list.argtypes = [u32, vp, u32, vp, u32, C.POINTER(u32)]
    L.bsg_survivor_rows_size.argtypes = [vp, vp, u32, u64, C.POINTER(u64), C.POINTER(u64)]
    L.bsg_survivor_rows_list.argtypes = [vp, vp, u32, u64, vp, vp, u32, u32, vp, u32, C.POINTER(u32)]
    L.bsg_survivor_rows_list_packed.argtypes = [vp, vp, u32, u64, vp, vp, u32, u32, vp, u32, C.POINTER(u32)]
    L.bsg_survivor_list.argtypes = [vp, u32, vp, u32, C.POINTER(u32)]
    L.bsg_query.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp, u32, vp]
    L.bsg_lab_query_cpu.argtypes = [vp, vp, i32]
    L.bsg_query_stats_read.argtypes = [vp, C.POINTER(QueryStats), i32]
    L.bsg_timing_read.argtypes = [vp, C.POINTER(Timing), i32]
    L.bsg_set_timed_stride.argtypes = [vp, u32]
    L.bsg_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bsg_or_reduce.argtypes = [vp, u64, u32, vp, u64]
    L.bsg_or_words_dev.argtypes = [vp, vp, vp, u64, u32]
    L.bsg_or_reduce_dev.argtypes = [vp, u64, u32, vp, u64]
    L.bsg_last_or_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsg_comm_unique_id.argtypes = [vp]
    L.bsg_comm_init.argtypes = [vp, vp, i32, i32]
    L.bsg_comm_destroy.argtypes = [vp]
    L.bsg_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.bsg_or_allreduce.argtypes = [vp, u64, u32, vp, u64]
    L.bsg_or_allreduce_dev.argtypes = [vp, vp, u64]
    L.bsg_ingest_rows.argtypes = [vp, vp, vp, u32, vp, u32, vp, u32, vp, u32, C.POINTER(u64)]
    L.bsg_ingest_fallback_rows.argtypes = [vp, u64, vp, u32, C.POINTER(u32)]
    L.bsg_ingest_add_entries.argtypes = [vp, u64, vp, vp, u32, vp, vp]
    L.bsg_ingest_finish.argtypes = [vp, u64, vp, vp]
    L.bsg_ingest_build.argtypes = [vp, u64, vp, vp, u64]
    L.bsg_ingest_stats_read.argtypes = [vp, u64, C.POINTER(IngestStats)]
    L.bsg_ingest_free.argtypes = [vp, u64]
    L.bsg_ingest_build_sections.argtypes = [vp, u64, vp, vp, u64, vp, C.POINTER(u64), C.POINTER(u64)]
    L.bsg_sections_size.argtypes = [vp, u32, C.POINTER(u64)]
    L.bsg_build_sections.argtypes = [vp, vp, vp, u32, vp, vp, u32, u64, vp, u64, vp]
    L.bsg_last_encode_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsg_match_rows.argtypes = [vp, vp, vp, u32, vp, vp, vp, u32, vp, u32, vp, vp, u32, C.POINTER(u32)]
    L.bsg_last_match_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsg_pinned_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.bsg_pinned_free.argtypes = [vp, vp]
    L.bsg_host_register.argtypes = [vp, vp, u64]
    L.bsg_host_unregister.argtypes = [vp, vp]
    for name in EXPORTS + LAB_EXPORTS:
        if name != "bsg_last_error":
            getattr(L, name).restype = i32
    _lib = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data if a.size else None
    return a
