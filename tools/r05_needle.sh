#!/bin/bash
# the many-term probe (k_probe_terms_many) on the needle batch (4 054 distinct field::token terms): kernel time per 64-arena launch
python bench.py --workload needle --cpu-budget 0 --c4-files 0 --ingest-blocks 0 --no-decode --or-union 0 --no-q1 --no-single --no-big-filters --no-concurrent --scaled 0 --steps 128 --warmup 64 --group 64 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); r=o['roofline']; k=r['all'][r['kernel']]
print('%s: %.1f us per launch of %.0f arenas, %.2f TB/s algorithmic = frac %.3f (%d samples); step %.2f us' % (r['kernel'], k['kernel_ms']*1e3, k['arenas_per_launch'], k['achieved']/1e3, k['frac'], k['samples'], o['ms_per_step']*1e3))"
