#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the standalone device-ingest run.  Usage: tools/profile_ingest_trace.sh <tag> [n_blocks]
set -u
TAG=$1; NB=${2:-40}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o ing -- python $REPO/tools/ingest_prof.py $NB 10000 3 1 > $OUT/stats.log 2>&1
cd $REPO
grep "^rep\|match:" $OUT/stats.log
python - <<PY
import glob, sqlite3
for d in glob.glob("$OUT/stats/**/*.db", recursive=True):
    c = sqlite3.connect(d)
    print("== rocprofv3 --kernel-trace --stats : top_kernels (name, calls, total us, avg us, %)")
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        print("  %-34s calls %5d  total %10.1f us  avg %10.3f us  %5.1f%%" % (name.split("(")[0].replace("bsg::", ""), calls, total, avg, pct))
    print("== launch shapes (grid_x, grid_y, workgroup_x, lds, vgprs, sgprs)")
    for n in [r[0] for r in c.execute("select distinct name from kernels where name like 'bsg::k_ingest%' or name like 'bsg::k_build_sets%' or name like 'bsg::k_match%'")]:
        print("  %-34s %s" % (n.split("(")[0].replace("bsg::", ""), c.execute("select grid_x, grid_y, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels where name = ? order by start desc limit 1", (n,)).fetchone()))
PY
