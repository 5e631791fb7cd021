"""The synchronisation of the bsg_query combiner on the CPU (no GPU): tests/combiner_sync_check.cpp instantiates
bloomsearch_amd/csrc/host/combiner_sync.hpp — the code the library itself uses: lock-free stacks of waiting calls, the collector role
and cycle slots, futex waits, the wake-up tree — over a stand-in cycle and drives it from many threads, under ThreadSanitizer where the
toolchain has it.  Every call must come back served exactly once with its own answer, calls must really share cycles, the gate must end
idle, and nothing may deadlock (the program carries a watchdog)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "combiner_sync_check.cpp")
INC = os.path.join(ROOT, "bloomsearch_amd", "csrc", "host")


def _build(tmp_path, tsan):
    exe = tmp_path / ("combiner_sync_check_tsan" if tsan else "combiner_sync_check")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", INC, "-o", str(exe), SRC] + (["-fsanitize=thread"] if tsan else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    return (exe if r.returncode == 0 else None), r.stderr


def _run(exe, args, env=None):
    r = subprocess.run([str(exe)] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=env)
    fields = dict(zip(r.stdout.split()[::2], r.stdout.split()[1::2])) if r.stdout else {}
    return r, fields


@pytest.mark.parametrize("threads,per,inflight,spin,work", [(1, 300, 2, 60, 2000), (4, 1500, 1, 0, 3000), (16, 1500, 2, 60, 5000),
                                                            (48, 600, 3, 0, 20000), (96, 300, 2, 5, 30000)])
def test_every_call_is_served_once_with_its_own_answer(tmp_path, threads, per, inflight, spin, work):
    exe, err = _build(tmp_path, tsan=False)
    assert exe is not None, err
    r, f = _run(exe, [threads, per, inflight, spin, work])
    assert r.returncode == 0, r.stdout + r.stderr
    assert int(f["calls"]) == threads * per and int(f["bad"]) == 0 and int(f["idle"]) == 1
    assert int(f["inflight_max"]) <= inflight + 1
    if threads >= 16:
        assert int(f["max_cycle"]) > 1 and int(f["cycles"]) < int(f["calls"]), f      # calls really shared cycles


def test_under_thread_sanitizer(tmp_path):
    exe, err = _build(tmp_path, tsan=True)
    if exe is None:
        pytest.skip("this toolchain has no ThreadSanitizer runtime: %s" % err.strip().splitlines()[-1:])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    for args in ([8, 400, 2, 20, 5000], [24, 200, 3, 0, 20000]):
        r, f = _run(exe, args, env)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, r.stdout + r.stderr
        assert int(f["bad"]) == 0 and int(f["idle"]) == 1
