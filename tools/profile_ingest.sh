#!/bin/bash
# Runs on the GPU box: SQ counter passes for the ingest kernels.  Usage: tools/profile_ingest.sh <tag> [n_blocks] [sets...]
set -u
TAG=$1; NB=${2:-20}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY")
if [ "${3:-}" = "full" ]; then SETS+=("SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "FETCH_SIZE WRITE_SIZE"); fi
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o ing -- python $REPO/tools/ingest_prof.py $NB 10000 1 > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import glob, sqlite3
for d in sorted(glob.glob("$OUT/p*/**/*.db", recursive=True)):
    c = sqlite3.connect(d)
    try:
        rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like 'bsg::k_ingest%' or kernel_name like 'bsg::k_build_sets%' group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(d, e); continue
    for k, n, v, cnt in rows:
        print("%-22s %-24s %16.0f  (%d dispatches)" % (k.split("(")[0].replace("bsg::", ""), n, v, cnt))
PY
