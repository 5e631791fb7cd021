#!/bin/bash
# SQ / TCP / TCC counters of the combiner's job kernel (GPU box, via gpurun): one rocprofv3 --pmc pass per counter set around
# tools/conc_lab.py.  Usage: CASES=2 TS=64 HOT=0 bash tools/conc_pmc.sh <tag>  -> gpurun_out/conc_pmc_<tag>.txt
set -u
TAG=${1:-jobs}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_conc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o b -- python $REPO/tools/conc_lab.py 0.15 2 > $OUT/p$i.log 2>&1
done
cd $REPO
python - "$OUT" > gpurun_out/conc_pmc_$TAG.txt 2>&1 <<'PY'
# every counter set is a run of its own with cycles of its own sizes: a counter is normalised by the workgroups of the dispatches IT
# was collected on (sum of values / sum of workgroups), never by another pass's wave count
import glob, os, sqlite3, sys
root = sys.argv[1]
vals = {}
for db in sorted(glob.glob(os.path.join(root, "p*", "**", "*.db"), recursive=True)):
    d = sqlite3.connect(db)
    try:
        rows = list(d.execute("select kernel_name, counter_name, sum(value), sum((grid_size_x / workgroup_size_x) * (grid_size_y / workgroup_size_y) * (grid_size_z / workgroup_size_z)), count(*) from counters_collection "
                              "where kernel_name like 'bsg::k_query%' group by kernel_name, counter_name"))
    except Exception as exc:
        print("db %s: %r" % (db, exc)); continue
    for name, ctr, v, wgs, n in rows:
        vals.setdefault(name.split("(")[0].replace("bsg::", ""), {})[ctr] = (v, wgs, n)
for k, c in sorted(vals.items()):
    print("== %s (a workgroup = 4 waves = 64 blocks of one (call, arena) pair)" % k)
    for ctr, (v, wgs, n) in sorted(c.items()):
        print("   %-32s %12.1f per workgroup   (%d dispatches, %.0f workgroups each on average)" % (ctr, v / max(wgs, 1), n, wgs / max(n, 1)))
PY
for f in $OUT/p*.log; do grep -i "error\|invalid\|not found" $f | head -2 | cut -c1-200; done >> gpurun_out/conc_pmc_$TAG.txt
rm -rf $OUT
cat gpurun_out/conc_pmc_$TAG.txt
