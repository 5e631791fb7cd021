#!/bin/bash
# One GPU-box pass that regenerates what profiles/ holds for a round: rocprofv3 stats (+ PMC traffic) for the C2, C4-shape and
# many-term workloads, the SQ counters of the many-term kernel, the driver-shaped and default bench lines, the round's labs.
# Usage (via gpurun): bash tools/final_round.sh r05        (kernels a round did not touch keep the earlier rounds' files)
set -u
R=${1:-r05}
mkdir -p gpurun_out gpurun_out/keep
bash tools/profile.sh ${R}_c2 > /dev/null
NO_PMC=1 bash tools/profile.sh ${R}_c4 --workload c4 > /dev/null
NO_PMC=1 bash tools/profile.sh ${R}_needle --workload needle > /dev/null
bash tools/profile_pmc.sh ${R}_needle --workload needle > /dev/null
# the driver's own command under the kernel trace: the 20-arena launch shape of its timed region next to roofline.kernel_ms
EXACT_ARGS="--steps 20 --warmup 5 --cpu-budget 0" NO_PMC=1 bash tools/profile.sh ${R}_driver > /dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_shape.json 2> gpurun_out/${R}_bench_driver_shape.err
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python bench.py --workload needle --cpu-budget 0 --c4-files 0 --ingest-blocks 0 --no-decode --or-union 0 --no-concurrent > gpurun_out/${R}_bench_needle.json 2>/dev/null
python bench.py --workload c4 --cpu-budget 6 --c4-files 0 --ingest-blocks 0 --no-decode --or-union 0 --no-concurrent > gpurun_out/${R}_bench_c4.json 2>/dev/null
# round 5 labs: concurrent bsg_query callers (alone vs combined, collector phases), the step's tail split and the per-rank shard
# sizes of the strong-scaling leg on one GPU (what N = 2 / 4 / 8 ranks each see), the section codec, the N > 1 host paths
HOT=8 python tools/conc_lab.py 0.5 2 > gpurun_out/${R}_conc_lab.txt 2>&1
# the job kernel of the combiner (k_query_jobs): durations by launch shape and the SQ / TCP / TCC counters of long lists (no hot arenas)
CASES=2 TS=16,64 HOT=0 bash tools/conc_prof.sh jobs > /dev/null 2>&1
CASES=2 TS=64 HOT=0 bash tools/conc_pmc.sh jobs > /dev/null 2>&1
cp gpurun_out/conc_prof_jobs.txt gpurun_out/keep/${R}_jobs_rocprofv3.txt; cp gpurun_out/conc_pmc_jobs.txt gpurun_out/keep/${R}_jobs_pmc.txt
bash tools/r05_step.sh > gpurun_out/${R}_step.txt 2>&1
grep "^tail split 0%" gpurun_out/${R}_step.txt | sed "s/^tail split 0%: //" > gpurun_out/${R}_c4_shard_sweep.txt
python tools/decode_lab.py 2>&1 | tail -2 > gpurun_out/${R}_decode_lab_run.txt
bash tools/r05_needle.sh > gpurun_out/${R}_needle_run.txt 2>&1
BSG_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --ingest-blocks 0 --no-decode --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --no-concurrent > gpurun_out/${R}_bench_gpus2_shared_gpu.json 2> gpurun_out/${R}_bench_gpus2_shared_gpu.err
BSG_BENCH_MULTI_CTX=8 python bench.py --steps 20 --warmup 5 --ingest-blocks 0 --no-decode --or-union 0 --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --no-concurrent --c4-files 0 --samples 2 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); print(json.dumps(o['multi_device_context'], indent=1))" > gpurun_out/${R}_multi_device_context.json
# the fuzzers' closing sweep (bounded: ~3 minutes in all); each line is the tool's own last line
(echo "== tools/fuzz_probe.py 12000 400"; timeout 300 python tools/fuzz_probe.py 12000 400 2>&1 | tail -1
 echo "== tools/fuzz_sections.py 12000 400"; timeout 120 python tools/fuzz_sections.py 12000 400 2>&1 | tail -1
 echo "== tools/fuzz_ingest_layout.py 1200 100"; timeout 120 python tools/fuzz_ingest_layout.py 1200 100 2>&1 | tail -1
 echo "== tools/fuzz_build.py 1200 100"; timeout 100 python tools/fuzz_build.py 1200 100 2>&1 | tail -1
 echo "== tools/fuzz_walker.py 1200 30"; timeout 120 python tools/fuzz_walker.py 1200 30 2>&1 | tail -1) > gpurun_out/${R}_fuzz.txt 2>&1
# what to keep: the summaries and the bench lines (the rocprofv3 databases stay in gpurun_out/)
cp gpurun_out/${R}_fuzz.txt gpurun_out/${R}_conc_lab.txt gpurun_out/${R}_step.txt gpurun_out/${R}_c4_shard_sweep.txt gpurun_out/${R}_decode_lab_run.txt gpurun_out/${R}_needle_run.txt \
   gpurun_out/${R}_bench_*.json gpurun_out/${R}_multi_device_context.json gpurun_out/keep/ 2>/dev/null
cp gpurun_out/prof_${R}_c2/summary.txt gpurun_out/keep/${R}_probe_c2_rocprofv3.txt
cp gpurun_out/prof_${R}_c4/summary.txt gpurun_out/keep/${R}_probe_c4_rocprofv3.txt
cp gpurun_out/prof_${R}_needle/summary.txt gpurun_out/keep/${R}_probe_needle_rocprofv3.txt
cp gpurun_out/prof_${R}_driver/summary.txt gpurun_out/keep/${R}_bench_driver_shape_rocprofv3.txt
cp gpurun_out/prof_${R}_c2/traffic.json gpurun_out/keep/${R}_traffic.json
cp gpurun_out/pmc_${R}_needle/summary.txt gpurun_out/keep/${R}_needle_pmc.txt
# gpurun merges at most 64 MiB back: the rocprofv3 databases (tens of MB each) stay on the box, the summaries travel
rm -rf gpurun_out/prof_${R}_* gpurun_out/pmc_${R}_*
ls -la gpurun_out/keep
