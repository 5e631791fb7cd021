"""Build libbloomgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m bloomsearch_amd.build [--force]

The .so lives next to the sources (bloomsearch_amd/csrc/libbloomgpu.so) so it
travels with the tree; it is git-ignored, not pip-installed.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(CSRC, "libbloomgpu.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "host", "*.cpp")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(CSRC, "host", "*.h*")) + glob.glob(os.path.join(CSRC, "host", "*.inc")) + \
        glob.glob(os.path.join(INCLUDE, "*.h"))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function", *os.environ.get("BSG_EXTRA_CXXFLAGS", "").split(),
           "-I", INCLUDE, "-I", CSRC, "-I", os.path.join(CSRC, "host"), "-o", LIB + ".tmp"] + sources() + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
