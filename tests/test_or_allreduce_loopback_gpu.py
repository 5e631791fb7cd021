"""The world > 1 schedule of bsg_or_allreduce (csrc/comm_api.inc: slice j of every partial to rank j, OR, in-place
all-gather) executed on ONE GPU: BSG_RCCL_LIBRARY binds the ten communicator symbols from tests/loopback_ccl.cpp, a test
double whose ranks are threads of one process.  Real RCCL runs the same calls in bench.py under torchrun (world = number
of GPUs) and at world 1 in test_configs_gpu.py; this test is what checks the slice / inbox / padding arithmetic for
worlds of 2, 3, 4 and 8 ranks and for contexts of several entries (BASELINE configs[4])."""
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libloopback_ccl.so")


def build_loopback():
    src = os.path.join(HERE, "loopback_ccl.cpp")
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(src):
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run([hipcc, "-O1", "-std=c++17", "-shared", "-fPIC", "-o", SO, src], check=True)
    return SO


def test_the_loopback_library_builds_and_exports_what_comm_api_binds():
    so = build_loopback()
    import ctypes
    lib = ctypes.CDLL(so)
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclAllGather", "ncclSend", "ncclRecv",
                "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString", "ncclCommCount", "ncclCommUserRank"):
        assert hasattr(lib, sym), sym
    # and comm_api.inc binds exactly these
    text = open(os.path.join(HERE, "..", "bloomsearch_amd", "csrc", "comm_api.inc")).read()
    import re
    assert sorted(set(re.findall(r'BSG_SYM\(\w+, "(\w+)"\)', text))) == sorted(
        ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclAllGather", "ncclSend", "ncclRecv",
         "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"])


@pytest.mark.gpu
def test_or_allreduce_at_worlds_of_2_3_4_8_ranks_and_on_contexts_of_several_entries():
    root = os.path.join(HERE, "..")
    env = dict(os.environ, BSG_RCCL_LIBRARY=build_loopback(), PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py")], env=env, cwd=os.path.join(HERE, ".."),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "loopback or_allreduce: ok" in r.stdout
    assert r.stdout.count(": ok") == 13, r.stdout                      # 7 worlds of threads + 5 contexts + the summary
    # the check has teeth: an exchange that loses the last rank's slices must be noticed
    bad = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py")], env=dict(env, LOOPBACK_CCL_BREAK="1"), cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "words differ" in bad.stdout, bad.stdout[-3000:] + bad.stderr[-3000:]


@pytest.mark.gpu
def test_c5_ten_thousand_fixed_geometry_filters_over_a_world_of_8_equal_the_oracle_build_of_the_union():
    """BASELINE configs[4] at its stated 10 000 block filters, 1 250 per rank on 8 ranks (threads over the loopback double on this
    box's one GPU): local OR + the library's exchange == the oracle's build of the union at the fixed geometry."""
    root = os.path.join(HERE, "..")
    env = dict(os.environ, BSG_RCCL_LIBRARY=build_loopback(), PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_worker.py"), "c5"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "10000 fixed-geometry filters" in r.stdout and ": ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_a_library_name_that_does_not_load_is_an_error_not_a_fallback():
    code = ("from bloomsearch_amd.gpu import Context, BloomGpuError\n"
            "try:\n    Context.comm_unique_id()\nexcept BloomGpuError as e:\n    print('refused:', e)\n")
    env = dict(os.environ, BSG_RCCL_LIBRARY="/nonexistent/libccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.join(HERE, ".."), capture_output=True, text=True, timeout=300)
    assert "refused:" in r.stdout and "did not load" in r.stdout, r.stdout + r.stderr
