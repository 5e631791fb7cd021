"""bench.py as the driver calls it.

`python bench.py --gpus N ...` WITHOUT a launcher around it must start its own N ranks (torch.distributed.run, one rank per
GPU) and print ONE JSON line whose headline at N > 1 is BASELINE configs[3] (C4, strong scaling) with the weak-scaling C2
run under `c2_weak`.  On a one-GPU box the ranks share device 0 (BSG_BENCH_SHARE_GPU=1: a functional check of every N > 1 host
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


def run_bench(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                       # the test itself may run under a launcher; bench.py must not think it does
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "bench.py failed (rc %d)\n--- stderr tail ---\n%s" % (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one JSON line, got %d:\n%s" % (len(lines), p.stdout[-2000:])
    return json.loads(lines[0]), p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks_and_reports_c4_strong_scaling():
    out, err = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5"] + SMALL, {"BSG_BENCH_SHARE_GPU": "1", "BSG_BENCH_RCCL_TIMEOUT": "60"})
    assert "re-executing as" in err
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["scaling"] == "strong" and out["config"]["workload"].startswith("C4")
    assert out["config"]["blocks_total"] == 256
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - out["config"]["probes_per_step"] * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-6 * out["value"]
    assert out["roofline"]["kernel"] in ("k_probe_eval", "k_probe_terms") and out["roofline"]["frac"] > 0
    per_rank = out["c4"]["per_rank"]
    assert [r["rank"] for r in per_rank] == [0, 1] and all(r["blocks"] == 128 and r["launches"] > 0 and r["kernel_ms"] > 0 for r in per_rank)
    c2 = out["c2_weak"]
    assert c2["scaling"] == "weak" and c2["value"] > 0 and c2["config"]["workload"].startswith("C2")
    assert out["c4"]["host_gather"]["rank0_view"].startswith("file 0: 2 ranks")
    # the N > 1 clock is the N = 1 clock: a rank's clock stops after its own synchronize and the job time is the MAX over ranks;
    # the closing barrier is timed BESIDE it (never less, and reported as its own field)
    ck = out["clock"]
    assert ck == out["c4"]["clock"]
    assert ck["ms_per_step_closing_barrier_inside"] >= out["ms_per_step"] > 0 and ck["closing_barrier_us"] >= 0
    assert 0 < ck["ms_per_step_this_rank"] <= out["ms_per_step"] * (1 + 1e-9)
    assert c2["clock"]["ms_per_step_closing_barrier_inside"] >= c2["ms_per_step"]
    # BASELINE configs[4]: 10 000 block filters in all, shared over the ranks
    assert out["or_reduce"]["filters_total"] == 10000 and out["or_reduce"]["filters_this_rank"] == 5000


@pytest.mark.gpu
def test_bench_gpus_1_keeps_c2_as_the_headline():
    out, err = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"] + SMALL)
    assert "re-executing" not in err
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["config"]["workload"].startswith("C2")
    assert out["c4"]["scaling"] == "strong" and out["c4"]["n_gpus"] == 1 and "c2_weak" not in out
    assert out["roofline"]["kernel"] in ("k_probe_eval", "k_probe_terms")
    # one rank: no closing barrier to time, both figures are the same clock
    assert out["clock"]["ms_per_step_closing_barrier_inside"] == out["ms_per_step"] and out["clock"]["closing_barrier_us"] == 0
