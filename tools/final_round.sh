#!/bin/bash
# One GPU-box pass that regenerates what profiles/ holds for a round: the driver-shaped and default bench lines (+ bench_legs.json), rocprofv3
# --kernel-trace --stats of the driver's exact command, the PMC traffic passes of the C2 probe, the round's labs, the fuzzers, the GPU suite.
# Usage (via gpurun): bash tools/final_round.sh r06      -> gpurun_out/keep/<tag>_*   (~10 minutes; the rocprofv3 databases stay on the box)
set -u
R=${1:-r06}
mkdir -p gpurun_out gpurun_out/keep
K=gpurun_out/keep
# 1. the line the driver records, twice: plain, and under the kernel trace (the SAME command: roofline.kernel_ms must agree with the trace's average)
python bench.py --gpus 1 --steps 20 --warmup 5 > $K/${R}_bench_driver_shape.json 2> gpurun_out/${R}_bench_driver_shape.err
cp bench_legs.json $K/${R}_bench_driver_shape_legs.json
tail -12 gpurun_out/${R}_bench_driver_shape.err > $K/${R}_bench_driver_shape_stderr_tail.txt
EXACT_ARGS="--gpus 1 --steps 20 --warmup 5" NO_PMC=1 bash tools/profile.sh ${R}_driver > /dev/null
cp gpurun_out/prof_${R}_driver/summary.txt $K/${R}_bench_driver_shape_rocprofv3.txt
# 2. PMC traffic of the C2 probe at 64 arenas per launch (separate --pmc passes, the guide's gfx950 correction) + kernel stats
bash tools/profile.sh ${R}_c2 > /dev/null
cp gpurun_out/prof_${R}_c2/summary.txt $K/${R}_probe_c2_rocprofv3.txt
cp gpurun_out/prof_${R}_c2/traffic.json $K/${R}_traffic.json
# 3. the default run (200 steps) and the C4 batch as the headline workload
python bench.py > $K/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
cp bench_legs.json $K/${R}_bench_default_legs.json
python bench.py --workload c4 --cpu-budget 6 --c4-files 0 --ingest-blocks 0 --no-decode --or-union 0 --no-concurrent --no-q1 --no-big-filters > $K/${R}_bench_c4.json 2>/dev/null
# 4. k_build under the SQ counters (tools/build_lab: 1 000 filters x 19 600 entries of 13 bytes)
bash tools/profile_build_pmc.sh ${R} > /dev/null 2>&1
cp gpurun_out/pmc_${R}/summary.txt $K/${R}_build_pmc.txt
# 5. labs: the strong-scaling shard sizes; concurrent callers (alone vs combined); the same beside a busy device; the file-arena cache
bash tools/shard_sweep.sh > $K/${R}_c4_shard_sweep.txt 2>&1
HOT=8 python tools/conc_lab.py 0.5 2 > $K/${R}_conc_lab.txt 2>&1
python tools/conc_busy.py > $K/${R}_conc_busy.txt 2>&1
python tools/cache_lab.py > $K/${R}_cache_lab.txt 2>&1
python tools/miss_lab.py > $K/${R}_miss_lab.txt 2>&1
# 6. the N > 1 host paths on one GPU (both ranks on device 0, gloo), and one context over 8 entries
BSG_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --ingest-blocks 0 --no-decode --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --no-concurrent > $K/${R}_bench_gpus2_shared_gpu.json 2> gpurun_out/${R}_bench_gpus2_shared_gpu.err
cp bench_legs.json $K/${R}_bench_gpus2_shared_gpu_legs.json
BSG_BENCH_MULTI_CTX=8 python bench.py --steps 20 --warmup 5 --ingest-blocks 0 --no-decode --or-union 0 --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --no-concurrent --c4-files 0 --samples 2 > /dev/null 2>&1
python -c "import json; print(json.dumps(json.load(open('bench_legs.json'))['multi_device_context'], indent=1))" > $K/${R}_multi_device_context.json
# 7. the fuzzers' closing sweep (bounded: ~3 minutes in all); each line is the tool's own last line
(echo "== tools/fuzz_probe.py 12000 400"; timeout 300 python tools/fuzz_probe.py 12000 400 2>&1 | tail -1
 echo "== tools/fuzz_sections.py 12000 400"; timeout 120 python tools/fuzz_sections.py 12000 400 2>&1 | tail -1
 echo "== tools/fuzz_ingest_layout.py 1200 100"; timeout 120 python tools/fuzz_ingest_layout.py 1200 100 2>&1 | tail -1
 echo "== tools/fuzz_build.py 1200 100"; timeout 100 python tools/fuzz_build.py 1200 100 2>&1 | tail -1
 echo "== tools/fuzz_walker.py 1200 30"; timeout 120 python tools/fuzz_walker.py 1200 30 2>&1 | tail -1) > $K/${R}_fuzz.txt 2>&1
# 8. the GPU suite and smoke at this commit
(python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|ERROR"; python __graft_entry__.py --smoke 2>&1 | grep -E "smoke|Error|error" | tail -2) > $K/${R}_gpu_suite.txt 2>&1
rm -rf gpurun_out/prof_${R}_* gpurun_out/pmc_${R}
ls -la $K
