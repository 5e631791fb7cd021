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


def test_struct_layouts_match_header():
    assert _lib.TERM_DTYPE.itemsize == 40 and _lib.DESC_DTYPE.itemsize == 24
    assert C.sizeof(_lib.Timing) == 32


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
