#!/bin/bash
# The host mirror (bloomsearch_amd/csrc/host/) under AddressSanitizer + UBSan, on a CPU-only box: host_api.cpp is built
# with g++ -fsanitize=address,undefined against generated stubs of the bsg_* entry points (nothing here touches a GPU),
# then the host-table tests and tools/fuzz_host.py run on it.  Usage: bash tools/asan_host.sh [seeds]
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/bsg_asan
mkdir -p $OUT
python3 - "$REPO" "$OUT" <<'PY'
import re, sys
repo, out = sys.argv[1:3]
h = open(repo + "/include/bloomgpu.h").read()
protos = re.findall(r"BSG_API\s+([^;{]*?\([^;{]*?\))\s*;", h, flags=re.S)
src = ['#include "bloomgpu.h"\nextern "C" {\n']
for p in protos:
    p = " ".join(p.split())
    ret = p[: p.index("(")].rsplit(" ", 1)[0].strip()
    body = '{ return "stub"; }' if ret == "const char *" or ret == "const char*" else ("{ return 0; }" if "*" in ret else ("{ }" if ret == "void" else "{ return -99; }"))
    if p.startswith("const char *"):
        body = '{ return "stub"; }'
    src.append(p + " " + body + "\n")
src.append("}\n")
open(out + "/stubs.cpp", "w").write("".join(src))
PY
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -shared -fPIC -I $REPO/include -I $REPO/bloomsearch_amd/csrc \
    -I $REPO/bloomsearch_amd/csrc/host -o $OUT/libbloomgpu_asan.so $OUT/stubs.cpp $REPO/bloomsearch_amd/csrc/host/host_api.cpp -lpthread
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export BSG_LAB_LIB=$OUT/libbloomgpu_asan.so
cd $REPO
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from bloomsearch_amd import _lib
_lib.LIB_PATH = os.environ["BSG_LAB_LIB"]
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "tests/test_host_tables.py", "-p", "no:cacheprovider"]))
PY
for s in $(seq 1 ${1:-3}); do python3 tools/fuzz_host.py $s 20000; done
