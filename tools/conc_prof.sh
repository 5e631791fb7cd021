#!/bin/bash
# Kernel durations of the combiner's dispatches (GPU box, via gpurun): rocprofv3 --kernel-trace --stats around tools/conc_lab.py.
# Usage: CASES=2 TS=16 bash tools/conc_prof.sh <tag>   -> gpurun_out/conc_prof_<tag>.txt  (CASES: 0 one arena, 1 one of 12, 2 ten of 12)
set -u
TAG=${1:-conc}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_conc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/tools/conc_lab.py 0.3 2 > $OUT/stats.log 2>&1
cd $REPO
(cat $OUT/stats.log | grep -v "^[WE]2026" | tr "|" "\n" | cut -c1-330; python tools/summarize_prof.py $OUT | grep -v "^== no database\|^== traffic"; python - $OUT <<'PY'
# the job kernel by list length: nanoseconds per workgroup (a workgroup = 64 blocks of one (call, arena) pair)
import glob, os, sqlite3, sys
for db in glob.glob(os.path.join(sys.argv[1], "stats", "**", "*.db"), recursive=True):
    d = sqlite3.connect(db)
    rows = list(d.execute("select grid_x / workgroup_x, duration from kernels where name like 'bsg::k_query_jobs%'"))
    print("== k_query_jobs / k_query_jobs_inline: duration by workgroups of the dispatch (%d dispatches)" % len(rows))
    for lo, hi in ((1, 128), (128, 512), (512, 2048), (2048, 8192), (8192, 1 << 30)):
        v = [(w, t) for w, t in rows if lo <= w < hi]
        if v:
            W, T = sum(w for w, _ in v), sum(t for _, t in v)
            print("   %5d <= workgroups < %-10d n %5d  mean %8.1f workgroups  mean %8.2f us  %6.1f ns per workgroup = %5.1f workgroups per us" % (lo, hi, len(v), W / len(v), T / len(v) / 1e3, T / W, W / T * 1e3))
PY
) > gpurun_out/conc_prof_$TAG.txt 2>&1
rm -rf $OUT
cat gpurun_out/conc_prof_$TAG.txt
