"""bench.py as the driver calls it.

`python bench.py --gpus N ...` WITHOUT a launcher around it must start its own N ranks (torch.distributed.run, one rank per
GPU) and print ONE compact JSON line (< 4 KB, strict JSON) whose headline at N > 1 is BASELINE configs[3] (C4, strong scaling); every leg's
full object — the weak-scaling C2 run under `c2_weak` among them — goes to bench_legs.json.  On a one-GPU box the ranks share device 0 (BSG_BENCH_SHARE_GPU=1: a functional check of every N > 1 host
path — sharding, shared-segment gather, per-rank statistics, rank-0 assembly; the numbers mean nothing and RCCL refuses two
ranks on one device, so the OR all-reduce leg reports its error in the line instead of a time)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL = ["--blocks", "128", "--rows-per-block", "1000", "--queries", "512", "--c4-files", "2", "--c4-blocks-per-file", "128",
         "--ingest-blocks", "0", "--scaled", "0", "--no-decode", "--cpu-budget", "0", "--or-union", "20000", "--no-q1", "--no-single",
         "--samples", "2", "--no-big-filters", "--no-concurrent"]


def fail_constant(name):
    raise ValueError("non-finite constant %s in the bench line" % name)


def run_bench(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                       # the test itself may run under a launcher; bench.py must not think it does
    env.update(env_extra or {})
    legs_path = os.path.join(ROOT, "bench_legs.json")
    if os.path.exists(legs_path):
        os.unlink(legs_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "bench.py failed (rc %d)\n--- stderr tail ---\n%s" % (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one JSON line, got %d:\n%s" % (len(lines), p.stdout[-2000:])
    # the driver's parser gave up on round 5's 25 KB line: the headline is compact, strict JSON (no NaN / Infinity)
    assert len(lines[0]) < 4096, len(lines[0])
    out = json.loads(lines[0], parse_constant=fail_constant)
    legs = json.load(open(legs_path), parse_constant=fail_constant)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "clock", "value_survivors_delivered_to_host"):
        assert key in out, key
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms", "samples"):
        assert key in out["roofline"], key
    assert "device-resident" in out["config"]["workload"]
    return out, legs, p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks_and_reports_c4_strong_scaling():
    out, legs, err = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5"] + SMALL, {"BSG_BENCH_SHARE_GPU": "1", "BSG_BENCH_RCCL_TIMEOUT": "60"})
    assert "re-executing as" in err
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["scaling"] == "strong" and out["config"]["workload"].startswith("C4")
    assert out["config"]["blocks_total"] == 256
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - out["config"]["probes_per_step"] / (out["ms_per_step"] * 1e-3)) < 1e-4 * out["value"]      # (6 significant digits in the line)
    assert out["roofline"]["kernel"] in ("k_probe_eval", "k_probe_terms") and out["roofline"]["frac"] > 0
    per_rank = legs["c4"]["per_rank"]
    assert [r["rank"] for r in per_rank] == [0, 1] and all(r["blocks"] == 128 and r["launches"] > 0 and r["kernel_ms"] > 0 for r in per_rank)
    c2 = legs["c2_weak"]
    assert c2["scaling"] == "weak" and c2["value"] > 0 and c2["config"]["workload"].startswith("C2")
    assert legs["c4"]["host_gather"]["rank0_view"].startswith("file 0: 2 ranks")
    # the contract's region: the closing barrier is INSIDE value / ms_per_step; the ranks' own clocks (closing barrier
    # outside) are reported beside it and are never longer
    ck = out["clock"]
    assert 0 < ck["ms_per_step_rank_clock"] <= out["ms_per_step"] * (1 + 1e-5) and ck["closing_barrier_us"] >= 0
    assert 0 < ck["ms_per_step_this_rank"] <= ck["ms_per_step_rank_clock"] * (1 + 1e-5)
    assert abs(legs["c4"]["ms_per_step"] - out["ms_per_step"]) < 1e-4 * out["ms_per_step"]
    assert c2["clock"]["ms_per_step_rank_clock"] <= c2["ms_per_step"] * (1 + 1e-5)
    # BASELINE configs[4]: 10 000 block filters in all, shared over the ranks
    assert legs["or_reduce"]["filters_total"] == 10000 and legs["or_reduce"]["filters_this_rank"] == 5000
    assert out["or_reduce"]["filters_this_rank"] == 5000


@pytest.mark.gpu
def test_bench_gpus_1_keeps_c2_as_the_headline():
    out, legs, err = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"] + SMALL)
    assert "re-executing" not in err
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["config"]["workload"].startswith("C2")
    assert legs["c4"]["scaling"] == "strong" and legs["c4"]["n_gpus"] == 1 and "c2_weak" not in legs
    assert out["c4"]["value"] > 0 and out["c4"]["blocks"] == 256
    assert out["roofline"]["kernel"] in ("k_probe_eval", "k_probe_terms") and out["roofline"]["frac"] > 0
    # C3 beside it: the warm median of several calls, the first call apart
    assert out["build"]["kernel"] == "k_build" and out["build"]["calls"] >= 6 and out["build"]["kernel_ms"] > 0 and out["build"]["first_call_ms"] > 0
    # one rank: no closing barrier to wait in, both clocks agree to the call overhead
    assert out["clock"]["closing_barrier_us"] < 50


@pytest.mark.gpu
def test_bench_line_as_the_driver_runs_it_is_compact_and_carries_roofline_and_cpu_baseline():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` — the driver's exact command, every leg on: one line under 4 KB of strict
    JSON with roofline.frac and cpu_baseline.value; cores = the CPUs the process may use, not the box's logical count."""
    out, legs, _ = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], timeout=1500)
    assert out["steps"] == 20 and out["warmup"] == 5
    assert 0 < out["roofline"]["frac"] <= 1 and out["roofline"]["kernel_ms"] > 0
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("port", "reference") and cb["unit"] == "probes/s"
    sys.path.insert(0, ROOT)
    from benchlib.common import effective_cpus
    assert (cb["cores"], cb["logical_cpus"]) == effective_cpus()
    for leg in ("build", "c4", "decode", "ingest", "or_reduce", "q1", "concurrent_queries", "roofline_scaled", "big_filters", "c2_probe"):
        assert leg in legs, leg
