#!/bin/bash
# SQ counters of the probe kernels for one bench workload, one rocprofv3 --pmc pass per counter set (no trace domains
# beyond --kernel-trace).  Usage: tools/profile_pmc.sh <tag> [bench args...]  -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 64 --warmup 64 --group 64 --samples 4 --cpu-budget 0 --no-check --no-decode --ingest-blocks 0 --or-union 0 --scaled 0 --no-q1 --no-single --no-big-filters --no-concurrent --c4-files 0 $*"
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o b -- python $REPO/bench.py $ARGS > $OUT/p$i.log 2>&1
done
cd $REPO
python - "$OUT" > $OUT/summary.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
root = sys.argv[1]
vals = {}
for db in sorted(glob.glob(os.path.join(root, "p*", "**", "*.db"), recursive=True)):
    d = sqlite3.connect(db)
    for name, ctr, z, v, n in d.execute("select kernel_name, counter_name, grid_size_z, avg(value), count(*) from counters_collection "
                                        "where kernel_name like 'bsg::k_probe%' group by kernel_name, counter_name, grid_size_z"):
        vals.setdefault((name.split("(")[0].replace("bsg::", ""), z), {})[ctr] = (v, n)
for (k, z), c in sorted(vals.items()):
    print("== %s, %d arenas per dispatch (%d dispatches)" % (k, z, max(n for _, n in c.values())))
    waves = c.get("SQ_WAVES", (0, 0))[0]
    for ctr, (v, n) in sorted(c.items()):
        print("   %-24s %16.0f%s" % (ctr, v, ("   per wave %10.1f" % (v / waves)) if waves and ctr != "SQ_WAVES" else ""))
PY
cat $OUT/summary.txt
