#!/bin/bash
# The strong-scaling shard on ONE GPU: the C4 leg (BASELINE configs[3]) at the blocks per file a rank of N = 1 / 2 / 4 / 8 holds
# (1 000 / 500 / 250 / 125), exactly 20 timed steps as the driver runs them.  Predicts the N-GPU speed-up before stragglers.
#   bash tools/shard_sweep.sh > gpurun_out/rNN_c4_shard_sweep.txt
COMMON="--steps 20 --warmup 5 --ingest-blocks 0 --no-decode --or-union 0 --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --no-concurrent --samples 0"
for bpf in 1000 500 250 125; do
  python bench.py $COMMON --c4-blocks-per-file $bpf > /dev/null 2>&1
  python - $bpf <<'PY'
import json, sys
bpf = int(sys.argv[1])
c = json.load(open("bench_legs.json"))["c4"]
print("%4d blocks per file held by this GPU (= a rank of N = %d): %.2f us per step device-resident, %.2f with survivor rows delivered to the host; kernels %s"
      % (bpf, 1000 // bpf, c["ms_per_step"] * 1e3, c["host_gather"]["rows"]["ms_per_step"] * 1e3,
         {k: (round(v["kernel_ms"] * 1e3, 1), v.get("samples"), v.get("arenas_per_launch")) for k, v in c["kernels"].items()}))
PY
done
