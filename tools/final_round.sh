#!/bin/bash
# One GPU-box pass that regenerates everything profiles/ holds for a round: rocprofv3 stats (+ PMC traffic) for the C2,
# C4-shape and many-term workloads, the SQ counters of the many-term kernel, the driver-shaped and default bench lines.
# Usage (via gpurun): bash tools/final_round.sh r03
set -u
R=${1:-r04}
mkdir -p gpurun_out gpurun_out/keep
bash tools/profile.sh ${R}_c2 > /dev/null
NO_PMC=1 bash tools/profile.sh ${R}_c4 --workload c4 > /dev/null
NO_PMC=1 bash tools/profile.sh ${R}_needle --workload needle > /dev/null
bash tools/profile_pmc.sh ${R}_needle --workload needle > /dev/null
# the driver's own command under the kernel trace: the 20-arena launch shape of its timed region next to roofline.kernel_ms
EXACT_ARGS="--steps 20 --warmup 5 --cpu-budget 0" NO_PMC=1 bash tools/profile.sh ${R}_driver > /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_shape.json 2> gpurun_out/${R}_bench_driver_shape.err
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python bench.py --workload needle --cpu-budget 0 --c4-files 0 --ingest-blocks 0 --no-decode --or-union 0 > gpurun_out/${R}_bench_needle.json 2>/dev/null
python bench.py --workload c4 --cpu-budget 6 --c4-files 0 --ingest-blocks 0 --no-decode --or-union 0 > gpurun_out/${R}_bench_c4.json 2>/dev/null
bash tools/profile_ingest_trace.sh ${R}_ingest 300 > gpurun_out/${R}_ingest_rocprofv3.txt 2>&1
bash tools/profile_build_pmc.sh ${R}_build > /dev/null 2>&1
bash tools/profile_ingest_traffic.sh ${R}_ingest_traffic 300 > gpurun_out/${R}_ingest_traffic.txt 2>&1
tools/or_lab 1000 44976 20 > gpurun_out/${R}_or_lab.txt 2>&1
NB=1000 bash tools/union_ab.sh > gpurun_out/${R}_union_ab.txt 2>&1
# round 4 labs: the folded dispatch against two dispatches (20 / 128 arenas, C2 and the C4 batch), the many-term kernel, the section
# codec, the per-rank shard sizes of the strong-scaling leg on one GPU (what N = 2 / 4 / 8 ranks each see), and the N > 1 host paths
for a in "20" "128" "20 c4" "64 needle"; do echo "== tools/fold_lab.py $a"; python tools/fold_lab.py $a 2>&1 | grep -v "^$"; done > gpurun_out/${R}_fold_lab.txt 2>&1
python tools/decode_lab.py > gpurun_out/${R}_decode_lab.txt 2>&1
for bpf in 1000 500 250 125; do python bench.py --steps 20 --warmup 5 --ingest-blocks 0 --no-decode --or-union 0 --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --c4-blocks-per-file $bpf 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); c=o['c4']; print('$bpf blocks per file held by this GPU (= N = %d ranks): %.2f us per step bare, %.2f with dispatch timestamps; rows to host %.2f; kernels %s' % (1000 // $bpf, c['ms_per_step']*1e3, c['ms_per_step_with_dispatch_timestamps']*1e3, c['host_gather']['rows']['ms_per_step']*1e3, {k:(round(v['kernel_ms']*1e3,1), v.get('arenas_per_launch')) for k,v in c['kernels'].items()}))"; done > gpurun_out/${R}_c4_shard_sweep.txt 2>&1
BSG_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --ingest-blocks 0 --no-decode --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters > gpurun_out/${R}_bench_gpus2_shared_gpu.json 2> gpurun_out/${R}_bench_gpus2_shared_gpu.err
# the fuzzers' closing sweep (bounded: ~3 minutes in all); each line is the tool's own last line
(echo "== tools/fuzz_probe.py 7000 600"; timeout 200 python tools/fuzz_probe.py 7000 600 2>&1 | tail -1
 echo "== tools/fuzz_sections.py 7000 400"; timeout 120 python tools/fuzz_sections.py 7000 400 2>&1 | tail -1
 echo "== tools/fuzz_ingest_layout.py 700 150"; timeout 120 python tools/fuzz_ingest_layout.py 700 150 2>&1 | tail -1
 echo "== tools/fuzz_build.py 700 150"; timeout 100 python tools/fuzz_build.py 700 150 2>&1 | tail -1
 echo "== tools/fuzz_walker.py 700 50"; timeout 120 python tools/fuzz_walker.py 700 50 2>&1 | tail -1) > gpurun_out/${R}_fuzz.txt 2>&1
cp gpurun_out/${R}_fuzz.txt gpurun_out/keep/ 2>/dev/null
cp gpurun_out/${R}_fold_lab.txt gpurun_out/${R}_decode_lab.txt gpurun_out/${R}_c4_shard_sweep.txt gpurun_out/${R}_bench_gpus2_shared_gpu.json gpurun_out/keep/ 2>/dev/null
# what to keep: the summaries and the bench lines (the rocprofv3 databases stay in gpurun_out/)
mkdir -p gpurun_out/keep
cp gpurun_out/prof_${R}_c2/summary.txt gpurun_out/keep/${R}_probe_c2_rocprofv3.txt
cp gpurun_out/prof_${R}_c4/summary.txt gpurun_out/keep/${R}_probe_c4_rocprofv3.txt
cp gpurun_out/prof_${R}_needle/summary.txt gpurun_out/keep/${R}_probe_needle_rocprofv3.txt
cp gpurun_out/prof_${R}_driver/summary.txt gpurun_out/keep/${R}_bench_driver_shape_rocprofv3.txt
cp gpurun_out/prof_${R}_c2/traffic.json gpurun_out/keep/${R}_traffic.json
cp gpurun_out/pmc_${R}_needle/summary.txt gpurun_out/keep/${R}_needle_pmc.txt
cp gpurun_out/${R}_ingest_rocprofv3.txt gpurun_out/${R}_ingest_traffic.txt gpurun_out/${R}_bench_*.json gpurun_out/${R}_or_lab.txt gpurun_out/${R}_union_ab.txt gpurun_out/keep/
cp gpurun_out/pmc_${R}_build/summary.txt gpurun_out/keep/${R}_build_pmc.txt
# gpurun merges at most 64 MiB back: the rocprofv3 databases (tens of MB each) stay on the box, the summaries travel
rm -rf gpurun_out/prof_${R}_* gpurun_out/pmc_${R}_*
ls -la gpurun_out/keep
