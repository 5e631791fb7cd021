#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{stats,fetch,write,summary.txt,traffic.json}
# Every probe dispatch of the run covers 64 arenas (steps = warmup = group = 64), so the per-kernel averages of
# `rocprofv3 --stats` describe one launch shape; tools/summarize_prof.py also breaks them down by grid.
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 64 --warmup 64 --group 64 --samples 16 --cpu-budget 0 --no-check --no-decode --ingest-blocks 0 --or-union 0 --scaled 0 --no-q1 --no-single --no-big-filters --no-concurrent --c4-files 0 $*"
# EXACT_ARGS="--steps 20 --warmup 5": profile exactly that bench command instead (the driver's), every leg included
if [ -n "${EXACT_ARGS:-}" ]; then ARGS="$EXACT_ARGS"; fi
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
if [ -z "${NO_PMC:-}" ]; then
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o bench -- python $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o bench -- python $REPO/bench.py $ARGS > $OUT/write.log 2>&1
fi
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
