#!/bin/bash
# The first run on a node with N > 1 MI355X (BASELINE configs[3], [4]).  Nothing here has ever seen two devices:
# the 1-GPU boxes of rounds 1-6 ran the N > 1 host paths with every rank / context entry on device 0.  One command, in order:
#   1. the scaling curve: bench.py --gpus 1 / 2 / 4 / 8 (as many as the node has), one process per GPU under torch.distributed.run,
#      the real RCCL bsg_or_allreduce in the closing leg with bsg_comm_info's world == N asserted by bench.py itself
#      (or_reduce.n_ranks_seen_by_rccl; a mismatch is reported as allreduce_error), BSG_BENCH_MULTI_CTX off;
#   2. the multi-device tests with REAL devices 0 .. N-1 (tests/helpers.py::device_ids picks them when torch sees enough GPUs):
#      peer copies, bsg_peer_access, per-device PCIe slices, sharded build / ingest / rows — until now (0,)*N aliases of one GPU;
#   3. the C4 strong-scaling table out of the four lines (value at N over c4.value at N = 1).
# Usage: bash tools/first_8gpu.sh [tag]      -> gpurun_out/<tag>_*.json / .err / .txt        (~10 minutes on 8 GPUs)
set -u
TAG=${1:-first8}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
unset BSG_BENCH_MULTI_CTX BSG_BENCH_SHARE_GPU BSG_TEST_ALIAS_DEVICES
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count() if torch.cuda.is_available() else 0)
PY
)
echo "[first_8gpu] $NGPU GPU(s) visible" | tee gpurun_out/${TAG}_summary.txt
if [ "$NGPU" -lt 2 ]; then
  echo "[first_8gpu] needs at least 2 GPUs: nothing to do that the 1-GPU rounds have not done" | tee -a gpurun_out/${TAG}_summary.txt
  exit 2
fi
python __graft_entry__.py > gpurun_out/${TAG}_build.log 2>&1 || { echo "[first_8gpu] build failed"; tail -5 gpurun_out/${TAG}_build.log; exit 1; }
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  # the driver's own command at every N (bench.py launches its N ranks itself when it finds no launcher around it)
  timeout 1500 python bench.py --gpus $N --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n$N.json 2> gpurun_out/${TAG}_bench_n$N.err
  echo "[first_8gpu] bench.py --gpus $N: rc $?" | tee -a gpurun_out/${TAG}_summary.txt
  cp bench_legs.json gpurun_out/${TAG}_bench_legs_n$N.json 2>/dev/null
done
python - "$TAG" <<'PY' | tee -a gpurun_out/${TAG}_summary.txt
import json, sys
tag = sys.argv[1]
base = None
for n in (1, 2, 4, 8):
    try:
        line = json.loads(open("gpurun_out/%s_bench_n%d.json" % (tag, n)).read())
        legs = json.load(open("gpurun_out/%s_bench_legs_n%d.json" % (tag, n)))
    except Exception as exc:
        continue
    c4 = line["c4"]["value"] if n == 1 else line["value"]          # C4 (strong scaling) is the headline at N > 1, a side object at N = 1
    base = base or c4
    orr = legs.get("or_reduce", {})
    print("N=%d  C4 %.3e probes/s (%.2fx of N=1)  ms/step %.4f  roofline.frac %.3f  | OR all-reduce: %s ranks seen by RCCL, %s ms%s"
          % (n, c4, c4 / base, line["ms_per_step"] if n > 1 else line["c4"]["ms_per_step"], line["roofline"]["frac"] or 0,
             orr.get("n_ranks_seen_by_rccl", "-"), orr.get("allreduce_ms", "-"), ("  ERROR " + orr["allreduce_error"]) if "allreduce_error" in orr else ""))
    if n > 1 and orr.get("n_ranks_seen_by_rccl") != n:
        print("     !! RCCL did not see %d ranks" % n)
    mdc = legs.get("multi_device_context") or {}
    if mdc:
        print("     one context over devices %s: %s; peer_access %s" % (mdc.get("devices"), mdc.get("error") or mdc.get("check"), mdc.get("peer_access")))
PY
# the multi-device tests on real devices
timeout 2400 python -m pytest tests -q -m gpu -k "multi_device or shard or sharded or rows_on_a_context or context_over or devices" > gpurun_out/${TAG}_multi_device_tests.txt 2>&1
echo "[first_8gpu] multi-device tests: rc $? ($(tail -1 gpurun_out/${TAG}_multi_device_tests.txt))" | tee -a gpurun_out/${TAG}_summary.txt
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_suite.txt 2>&1
echo "[first_8gpu] whole GPU suite: rc $? ($(tail -1 gpurun_out/${TAG}_gpu_suite.txt))" | tee -a gpurun_out/${TAG}_summary.txt
