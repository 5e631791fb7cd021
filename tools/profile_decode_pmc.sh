#!/bin/bash
# SQ counters of k_decode_sections (tools/decode_lab.py: 1 000 sections in one launch and in four), one rocprofv3 --pmc pass per
# counter set.  Usage: tools/profile_decode_pmc.sh <tag>  -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o b -- python $REPO/tools/decode_lab.py > $OUT/p$i.log 2>&1
done
cd $REPO
python - "$OUT" > $OUT/summary.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
root = sys.argv[1]
vals = {}
for db in sorted(glob.glob(os.path.join(root, "p*", "**", "*.db"), recursive=True)):
    d = sqlite3.connect(db)
    try:
        rows = list(d.execute("select kernel_name, counter_name, grid_size_y, avg(value), count(*) from counters_collection "
                              "where kernel_name like 'bsg::k_decode%' group by kernel_name, counter_name, grid_size_y"))
    except Exception as exc:
        print("db %s: %r" % (db, exc)); continue
    for name, ctr, y, v, n in rows:
        vals.setdefault((name.split("(")[0].replace("bsg::", ""), y), {})[ctr] = (v, n)
for (k, y), c in sorted(vals.items()):
    print("== %s, %d sections per dispatch (%d dispatches)" % (k, y, max(n for _, n in c.values())))
    waves = c.get("SQ_WAVES", (0, 0))[0]
    for ctr, (v, n) in sorted(c.items()):
        print("   %-28s %16.0f%s" % (ctr, v, ("   per wave %10.1f" % (v / waves)) if waves and ctr != "SQ_WAVES" else ""))
PY
cat $OUT/summary.txt
