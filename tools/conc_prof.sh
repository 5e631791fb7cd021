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
(cat $OUT/stats.log | tr "|" "\n" | cut -c1-330; python tools/summarize_prof.py $OUT) > gpurun_out/conc_prof_$TAG.txt 2>&1
rm -rf $OUT
cat gpurun_out/conc_prof_$TAG.txt
