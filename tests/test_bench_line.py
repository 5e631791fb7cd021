"""bench.py's one line of stdout, without a GPU: it must stay under the size the driver's parser took in rounds 1-4 (round 5's 25 KB line
came back `parsed: null`), be strict JSON (NaN / inf -> null) and keep the contract's keys whatever the legs hold."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from benchlib import common as C  # noqa: E402


def fat(n):
    return {"k%d" % i: {"note": "x" * 200, "v": float(i) * 1.23456789e9} for i in range(n)}


def sample_headline():
    return {"metric": "block-bloom probes/sec", "value": 1854905447716.6006, "unit": "probes/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 0.006624596426263452, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2 probe, survivors device-resident: " + "w" * 200, "blocks_per_gpu": 1000, "queries": 4096, "distinct_terms": 29,
                       "probes_per_step": 12288000, "sharding": "round-robin blocks, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_probe_terms", "achieved": 6613.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.8266404320986,
                         "traffic": float("nan"), "traffic_source": "profiles/r06_traffic.json: " + "t" * 100, "algorithmic_bytes_per_launch": 703230000.0,
                         "kernel_ms": 0.1063, "samples": 17, "arenas_per_launch": 20.0, "copy_gbps": float("inf"), "frac_of_copy": None},
            "clock": {"ms_per_step_rank_clock": 0.0066, "ms_per_step_this_rank": 0.0066, "closing_barrier_us": 0.0, "barrier": "b" * 150},
            "value_survivors_delivered_to_host": 1.5e12, "survivors_note": "s" * 250,
            "cpu_baseline": {"value": 3.67e6, "unit": "probes/s", "cores": 16, "logical_cpus": 256, "kind": "port", "sample": "s" * 200}}


def legs():
    return {"build": dict(kernel="k_build", kernel_ms=0.69, first_call_ms=1.6, calls=6, algorithmic_bytes=716e6, achieved=1037.0, frac=0.13,
                          entries_per_s=5.6e10, warm_calls_ms=[0.7] * 5, workload="w" * 300),
            "c4": dict(value=5.3e12, ms_per_step=0.0614, steps=20, scaling="strong", blocks_total=10000, dominant_kernel="k_probe_terms",
                       kernels={"k_probe_terms": {"kernel_ms": 1.2, "frac": 0.8}}, host_gather={"rows": {"value": 5e12}}, **fat(30)),
            "roofline_scaled": dict(blocks=64000, kernel="k_probe_terms", kernel_ms=0.324, algorithmic_bytes_per_launch=2.25e9, achieved=6940.0, frac=0.867),
            "or_reduce": dict(kernel="k_or_reduce_blocks", kernel_ms=0.057, filters_this_rank=1000, achieved=6250.0, frac=0.78, **fat(10)),
            "concurrent_queries": fat(60), "q1": fat(20), "ingest": fat(20), "decode": fat(10), "big_filters": fat(10), "c2_probe": fat(10)}


def fail_constant(name):
    raise ValueError(name)


def test_headline_is_compact_strict_json_whatever_the_legs_hold():
    text = bench.headline(sample_headline(), legs())
    assert "\n" not in text and len(text) < bench.LINE_LIMIT <= 4096
    out = json.loads(text, parse_constant=fail_constant)
    assert out["roofline"]["traffic"] is None and out["roofline"]["copy_gbps"] is None          # NaN / inf -> null
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "clock", "value_survivors_delivered_to_host"):
        assert key in out, key
    assert out["build"]["kernel_ms"] == 0.69 and out["c4"]["blocks"] == 10000 and out["c4"]["frac"] == 0.8
    assert "concurrent_queries" not in out and "q1" not in out                                  # legs live in bench_legs.json
    assert abs(out["value"] - 1854905447716.6006) < 1e-5 * out["value"]


def test_headline_sheds_optional_objects_before_it_would_pass_the_limit():
    h = sample_headline()
    h["config"]["workload"] += "y" * 1500
    text = bench.headline(h, legs())
    assert len(text) < bench.LINE_LIMIT
    out = json.loads(text, parse_constant=fail_constant)
    assert "roofline" in out and "cpu_baseline" in out and "legs" not in out


def test_effective_cpus_respects_affinity_and_quota():
    n, logical = C.effective_cpus()
    assert 1 <= n <= logical == os.cpu_count()
    assert n <= len(os.sched_getaffinity(0))


def test_host_barrier_is_a_noop_alone():
    b = C.HostBarrier(0, 1)
    b.wait()
    b.close()
