#!/bin/bash
# Round 5: the step's serial tail and the strong-scaling shard (GPU box).  bash tools/r05_step.sh > gpurun_out/r05_step.txt
# (a) the driver's shape (20 steps of one C2 arena, one call) with the group's evaluation cut in two: --tail-split 0 / 40 / 50 / 60 / 70
# (b) the C4 leg at the shard sizes 1 / 2 / 4 / 8 ranks hold, tail split off and on
COMMON="--steps 20 --warmup 5 --ingest-blocks 0 --no-decode --or-union 0 --cpu-budget 0 --no-q1 --no-single --scaled 0 --no-big-filters --no-concurrent"
for ts in 0 40 50 60 70; do
  python bench.py $COMMON --c4-files 0 --tail-split $ts 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); r=o['roofline']
print('driver shape, tail split $ts%%: %.2f us per step bare (%.2f with dispatch timestamps); kernel %s %.1f us per launch of %s arenas = frac %.3f; eval %.1f us' % (o['ms_per_step']*1e3, o['ms_per_step_with_dispatch_timestamps']*1e3, r['kernel'], r['timed_region'][r['kernel']]['kernel_ms']*1e3, r['timed_region'][r['kernel']]['arenas_per_launch'], r['timed_region'][r['kernel']]['frac'], r['timed_region'].get('k_eval_programs',{}).get('kernel_ms',0)*1e3))"
done
for ts in 0 50; do for bpf in 1000 500 250 125; do
  python bench.py $COMMON --samples 0 --c4-blocks-per-file $bpf --tail-split $ts 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); c=o['c4']
print('tail split $ts%%: $bpf blocks per file held by this GPU (= N = %d ranks): %.2f us per step bare, %.2f with dispatch timestamps; rows to host %.2f; kernels %s' % (1000 // $bpf, c['ms_per_step']*1e3, c['ms_per_step_with_dispatch_timestamps']*1e3, c['host_gather']['rows']['ms_per_step']*1e3, {k:(round(v['kernel_ms']*1e3,1), v.get('samples'), v.get('arenas_per_launch')) for k,v in c['kernels'].items()}))"
done; done
