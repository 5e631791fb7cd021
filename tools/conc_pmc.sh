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
import glob, os, sqlite3, sys
root = sys.argv[1]
vals = {}
for db in sorted(glob.glob(os.path.join(root, "p*", "**", "*.db"), recursive=True)):
    d = sqlite3.connect(db)
    try:
        rows = list(d.execute("select kernel_name, counter_name, avg(value), avg(grid_size_x), count(*) from counters_collection "
                              "where kernel_name like 'bsg::k_query%' group by kernel_name, counter_name"))
    except Exception as exc:
        print("db %s: %r" % (db, exc)); continue
    for name, ctr, v, gx, n in rows:
        vals.setdefault(name.split("(")[0].replace("bsg::", ""), {})[ctr] = (v, gx, n)
for k, c in sorted(vals.items()):
    gx = max(g for _, g, _ in c.values())
    print("== %s: mean grid %.0f workgroups (%d dispatches)" % (k, gx / 256, max(n for _, _, n in c.values())))
    waves = c.get("SQ_WAVES", (0, 0, 0))[0]
    for ctr, (v, _, n) in sorted(c.items()):
        print("   %-32s %16.0f%s" % (ctr, v, ("   per wave %10.1f" % (v / waves)) if waves and ctr != "SQ_WAVES" else ""))
PY
for f in $OUT/p*.log; do grep -i "error\|invalid\|not found" $f | head -2 | cut -c1-200; done >> gpurun_out/conc_pmc_$TAG.txt
rm -rf $OUT
cat gpurun_out/conc_pmc_$TAG.txt
