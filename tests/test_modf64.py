"""x mod m through the fp64 pipe (kernels.hip.h: mod_f64, the build's and the many-term probe's route for 64 <= m <= 2^19) swept on the
HOST against the % operator: tests/modf64_check.hip includes the kernels' header and calls the very functions the kernels run (they are
__host__ __device__; one integer multiply-add and one IEEE fma on either side), no GPU needed.  ~1.4 x 10^8 values over ~7 000 moduli:
every m up to 4 096, the neighbours of every power of two up to 2^19, random ones; exact multiples of m and their neighbours at both
ends of the 64-bit range.  The GPU side of the same statement: tests/test_modulo_edges_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mod_f64_equals_the_remainder_operator(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = tmp_path / "modf64_check"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "modf64_check.hip")],
                   check=True, timeout=600)
    r = subprocess.run([str(exe), "5000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith(": ok"), r.stdout + r.stderr
