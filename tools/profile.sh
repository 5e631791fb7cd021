#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{stats,fetch,write}
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 200 --warmup 20 --cpu-budget 0 --no-check --no-decode --ingest-blocks 0 --or-union 0 $*"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o bench -- python $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o bench -- python $REPO/bench.py $ARGS > $OUT/write.log 2>&1
cd $REPO
find $OUT -type f | head -50
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
