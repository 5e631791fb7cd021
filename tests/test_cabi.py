"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, exports every symbol include/bloomgpu.h declares, and refuses to compute
without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from bloomsearch_amd import _lib, build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    B.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bloomgpu.h")).read()
    declared = set(re.findall(r"BSG_API\s+[\w\s\*]+?\b(bsg_\w+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    # the lab switches live in a header of their own: a binder of the contract never sees them
    lab = open(os.path.join(ROOT, "include", "bloomgpu_lab.h")).read()
    lab_declared = set(re.findall(r"BSG_API\s+[\w\s\*]+?\b(bsg_\w+)\s*\(", lab))
    assert lab_declared == set(_lib.LAB_EXPORTS) and not (lab_declared & declared)
    for name in lab_declared:
        assert hasattr(lib, name), name
    assert not re.search(r"bsg_set_lab|bsg_lab_|bsg_set_fuse_limit|bsg_set_timed_stride|bsg_set_spin_wait|bsg_set_gather_cost", hdr)


def test_struct_layouts_match_header():
    assert _lib.TERM_DTYPE.itemsize == 40 and _lib.DESC_DTYPE.itemsize == 24
    assert C.sizeof(_lib.Timing) == 112      # 10 fields of round 3 + the four k_probe_eval ones


def test_estimate_parameters_host_helper(lib):
    from bloomsearch_amd.gpu import estimate_parameters
    assert estimate_parameters(100, 0.01) == (959, 7)
    assert estimate_parameters(1, 0.001) == (15, 11)
    assert estimate_parameters(20000, 0.001) == (287552, 10)
    with pytest.raises(_lib.BloomGpuError):
        estimate_parameters(0, 0.01)


def test_no_cpu_fallback_without_gpu(lib):
    """Without a HIP device bsg_open must fail (BSG_E_NODEVICE) rather than run on the CPU."""
    if lib.bsg_device_count() > 0:
        pytest.skip("a GPU is visible here; the no-device path is exercised on CPU-only hosts")
    from bloomsearch_amd.gpu import Context
    with pytest.raises(_lib.BloomGpuError) as e:
        Context((0,))
    assert e.value.code == _lib.BSG_E_NODEVICE


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "bloomsearch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)


C_PROGRAM = r"""
/* A plain C99 caller of the boundary: no C++, no Python, no torch — what a cgo / JNI / FFI binding sees. */
#include <stdio.h>
#include <string.h>
#include "bloomgpu.h"
#include "bloomsearch_host.h"

/* TestEvaluateBloomFilters (bloom_tree_engine_test.go:357-442) from C, the way a cgo caller would run it: three (959, 7) filters
 * built and serialised on the device (bsg_build_sections), loaded back from the section bytes (bsg_arena_load_sections), the eight
 * expressions answered by ONE bsg_query call with the probed strings as they are, survivors listed by bsg_survivor_list. */
static int evaluate_fixture(bsg_ctx *ctx)
{
    static const char entries[] = "user.name" "user.age" "alice" "30" "user.name::alice" "user.age::30";
    const uint32_t off[7] = {0, 9, 17, 22, 24, 40, 52};
    const uint32_t fstart[4] = {0, 2, 4, 6};
    static const char probed[] = "user.name" "nonexistent.field" "alice" "user.name::alice";
    const uint32_t term_off[5] = {0, 9, 26, 31, 47};
    const uint32_t term_kinds[4] = {BSG_KIND_FIELD, BSG_KIND_FIELD, BSG_KIND_TOKEN, BSG_KIND_FIELD_TOKEN};
    /* nil query; field exists; field does not exist; token exists; field-token exists; OR one match; AND one mismatch; OR field / field-token */
    const uint32_t ops[14] = {BSG_OP(BSG_OP_TRUE, 0), BSG_OP(BSG_OP_TERM, 0), BSG_OP(BSG_OP_TERM, 1), BSG_OP(BSG_OP_TERM, 2), BSG_OP(BSG_OP_TERM, 3),
                              BSG_OP(BSG_OP_TERM, 1), BSG_OP(BSG_OP_TERM, 0), BSG_OP(BSG_OP_OR, 2),
                              BSG_OP(BSG_OP_TERM, 1), BSG_OP(BSG_OP_TERM, 0), BSG_OP(BSG_OP_AND, 2),
                              BSG_OP(BSG_OP_TERM, 1), BSG_OP(BSG_OP_TERM, 3), BSG_OP(BSG_OP_OR, 2)};
    const uint32_t prog_off[9] = {0, 1, 2, 3, 4, 5, 8, 11, 14};
    bsg_filter_desc d[3];
    uint8_t region[1024];
    uint64_t sec_off[2] = {0, 0}, arena = 0, total = 0, survivors[8];
    uint32_t blocks[4], n = 0;
    int32_t status[1] = {0};
    int i;
    memset(d, 0, sizeof d);
    for (i = 0; i < 3; ++i) { d[i].word_off = (uint64_t)i * 16; d[i].m = 959; d[i].k = 7; }
    if (bsg_sections_size(d, 1, &total) != BSG_OK || total > sizeof region) return 10;
    if (bsg_build_sections(ctx, (const uint8_t *)entries, off, 6, fstart, d, 3, 48, region, sizeof region, sec_off) != BSG_OK) return 11;
    if (sec_off[0] != 0 || sec_off[1] != total) return 12;
    printf("section=");
    for (i = 0; i < (int)total; ++i) printf("%02x", region[i]);
    printf("\n");
    if (bsg_arena_load_sections(ctx, region, total, sec_off, 1, status, &arena) != BSG_OK || status[0] != 0) return 13;
    if (bsg_query(ctx, &arena, 1, (const uint8_t *)probed, term_off, term_kinds, 4, ops, prog_off, 8, survivors) != BSG_OK) return 14;
    printf("verdicts=");
    for (i = 0; i < 8; ++i) printf("%d", (int)(survivors[i] & 1));
    printf("\n");
    if (bsg_survivor_list(&survivors[1], 1, blocks, 4, &n) != BSG_OK || n != 1 || blocks[0] != 0) return 15;
    if (bsg_survivor_list(&survivors[2], 1, blocks, 4, &n) != BSG_OK || n != 0) return 16;
    if (bsg_arena_free(ctx, arena) != BSG_OK) return 17;
    return 0;
}

int main(void)
{
    uint64_t m = 0, k = 0, total = 0;
    bsg_filter_desc d[3];
    bsg_term t;
    bsg_ingest_stats st;
    bsg_ctx *ctx = NULL;
    int32_t ids[1] = {0};
    int32_t rc;
    memset(d, 0, sizeof d); memset(&t, 0, sizeof t); memset(&st, 0, sizeof st);
    if (bsg_estimate_parameters(100, 0.01, &m, &k) != BSG_OK || m != 959 || k != 7) return 2;
    d[1].m = m; d[1].k = (uint32_t)k;
    if (bsg_sections_size(d, 1, &total) != BSG_OK || total != 1 + 4 + 24 + 8 * ((m + 63) / 64) + 4) return 3;
    if (sizeof(bsg_term) != 40 || sizeof(bsg_filter_desc) != 24) return 4;
    rc = bsg_open(ids, 1, &ctx);
    printf("devices=%d open=%d crc=%08x\n", (int)bsg_device_count(), (int)rc, (unsigned)bsh_crc32c((const uint8_t *)"123456789", 9));
    if (rc == BSG_OK) {
        uint64_t h[4] = {0, 0, 0, 0};
        const uint32_t off[2] = {0, 5};
        if (bsg_hash_entries(ctx, (const uint8_t *)"hello", off, 1, h) != BSG_OK) return 5;
        printf("hello=%016llx %016llx\n", (unsigned long long)h[0], (unsigned long long)h[1]);
        if ((rc = evaluate_fixture(ctx)) != 0) return rc;
        bsg_close(ctx);
    } else if (rc != BSG_E_NODEVICE) {
        return 6;
    }
    return 0;
}
"""


def run_c_caller(tmp_path):
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = tmp_path / "caller.c"
    src.write_text(C_PROGRAM)
    exe = tmp_path / "caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-l:libbloomgpu.so", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "crc=e3069283" in out.stdout.splitlines()[0]                      # CRC-32C check value
    return out.stdout


def test_plain_c99_program_links_against_the_boundary(lib, tmp_path):
    """include/*.h are C99 headers and libbloomgpu.so is an ordinary shared library: a C program compiled with gcc
    (-std=c99 -Wall -Werror -pedantic) links, runs, sizes sections on the host and — without a GPU — is refused by
    bsg_open with BSG_E_NODEVICE.  (tests/test_gpu_parity.py runs the same program where a GPU is present: there it also
    runs the reference's TestEvaluateBloomFilters fixture through bsg_build_sections / bsg_arena_load_sections / bsg_query.)"""
    out = run_c_caller(tmp_path)
    if lib.bsg_device_count() == 0:
        assert "open=-6" in out


def test_survivor_list_is_the_ascending_bit_positions():
    """bsg_survivor_list (host arithmetic, runs without a GPU): blockScanCandidate order = ascending block index, bits past
    the arena's end never count, a short list is reported (query_exec.go:321, 603)."""
    import numpy as np
    from bloomsearch_amd.gpu import survivor_list
    rng = np.random.default_rng(3)
    for n_blocks in (0, 1, 63, 64, 65, 130, 1000):
        bits = rng.integers(0, 2, size=n_blocks, dtype=np.uint8)
        padded = np.ones((n_blocks + 63) // 64 * 64, dtype=np.uint8)          # garbage past the end must be ignored
        padded[:n_blocks] = bits
        row = np.packbits(padded, bitorder="little").view(np.uint64) if n_blocks else np.zeros(0, dtype=np.uint64)
        assert survivor_list(row, n_blocks).tolist() == np.flatnonzero(bits).tolist()


def test_survivor_row_list_expands_every_tag_on_the_host():
    """bsg_survivor_row_list is host arithmetic (no device): a row of bsg_probe_many_rows, whatever its tag, becomes the ascending
    block indices blockScanCandidate walks (query_exec.go:321,603); rows_to_dense is its numpy twin."""
    import numpy as np
    from bloomsearch_amd.gpu import BloomGpuError, rows_to_dense, survivor_list, survivor_row_list
    n_blocks = 150
    G = (n_blocks + 63) // 64
    rng = np.random.default_rng(3)
    dense = rng.integers(0, 1 << 63, size=G, dtype=np.uint64)
    dense[-1] &= np.uint64((1 << (n_blocks & 63)) - 1)
    want_dense = survivor_list(dense, n_blocks)
    ids = np.asarray([0, 63, 64, 149], dtype=np.uint32)
    slot = np.zeros(G, dtype=np.uint64)
    slot.view(np.uint32)[: len(ids)] = ids
    cases = [((0 << 30) | 0, np.zeros(G, dtype=np.uint64), np.zeros(0, dtype=np.uint32)),
             ((1 << 30) | n_blocks, np.zeros(G, dtype=np.uint64), np.arange(n_blocks, dtype=np.uint32)),
             ((2 << 30) | len(ids), slot, ids),
             ((3 << 30) | len(want_dense), dense, want_dense)]
    for hdr, row, want in cases:
        assert np.array_equal(survivor_row_list(hdr, row, n_blocks), want), hdr >> 30
    back = rows_to_dense(np.asarray([c[0] for c in cases], dtype=np.uint32), np.concatenate([c[1] for c in cases]), n_blocks)
    for i, (_, _, want) in enumerate(cases):
        assert np.array_equal(survivor_list(back[i], n_blocks), want), i
    with pytest.raises(BloomGpuError):
        survivor_row_list((2 << 30) | (n_blocks + 1), slot, n_blocks)          # a count beyond the arena's blocks
    with pytest.raises(BloomGpuError):
        survivor_row_list((1 << 30) | 3, slot, n_blocks)                       # ALL with a count that is not the arena's
    with pytest.raises(BloomGpuError):
        survivor_row_list((3 << 30) | (len(want_dense) - 1), dense, n_blocks)  # DENSE whose words disagree with the header


def test_a_corrupt_list_header_is_rejected():
    """bsg_survivor_row_list never reads past a row's slot: a LIST header that counts more ids than the slot holds, or ids that are
    not ascending block numbers, is an error."""
    import ctypes as C
    import numpy as np
    from bloomsearch_amd import _lib
    L = _lib.load()
    n_blocks = 130                                            # G = 3: the slot holds 6 ids
    row = np.zeros(3, dtype=np.uint64)
    row.view(np.uint32)[:6] = [1, 5, 9, 64, 100, 129]
    out = np.zeros(n_blocks, dtype=np.uint32)
    n = C.c_uint32()
    assert L.bsg_survivor_row_list((2 << 30) | 6, row.ctypes.data, n_blocks, out.ctypes.data, n_blocks, C.byref(n)) == 0 and n.value == 6
    assert L.bsg_survivor_row_list((2 << 30) | 7, row.ctypes.data, n_blocks, out.ctypes.data, n_blocks, C.byref(n)) == _lib.BSG_E_INVALID
    row.view(np.uint32)[2] = 5                                # not ascending
    assert L.bsg_survivor_row_list((2 << 30) | 6, row.ctypes.data, n_blocks, out.ctypes.data, n_blocks, C.byref(n)) == _lib.BSG_E_INVALID
    row.view(np.uint32)[:6] = [1, 5, 9, 64, 100, 130]         # an id past the arena
    assert L.bsg_survivor_row_list((2 << 30) | 6, row.ctypes.data, n_blocks, out.ctypes.data, n_blocks, C.byref(n)) == _lib.BSG_E_INVALID


def test_gpu_test_modules_do_not_import_torch():
    """torch ships its own HIP / HSA runtime: once it is in the pytest process, /opt/rocm's librccl (bound by bsg_comm_init) finds an
    uninitialised HSA runtime and the real-RCCL test fails with "no ROCm-capable device is detected" (round 6).  The GPU tests talk
    to the device through libbloomgpu only; bench.py (torch first, the library bound to torch's librccl) runs in its own process."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "tests", "test_*_gpu.py")) + [os.path.join(ROOT, "tests", "helpers.py"), os.path.join(ROOT, "tests", "conftest.py")]:
        src = open(path).read()
        assert not re.search(r"^\s*(import torch|from torch)", src, re.M), path
