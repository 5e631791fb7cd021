#!/bin/bash
# SQ counters of k_build on the C3 block shape (tools/build_lab: F filters x E entries of L bytes, k = 10, m = 281 629), one
# rocprofv3 --pmc pass per counter set.  Usage: tools/profile_build_pmc.sh <tag> [F E L K]  -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="${*:-1000 19600 13 10}"
$REPO/tools/build_lab $ARGS > $OUT/unprofiled.txt 2>&1
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o b -- $REPO/tools/build_lab $ARGS > $OUT/p$i.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o b -- $REPO/tools/build_lab $ARGS > $OUT/pf.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o b -- $REPO/tools/build_lab $ARGS > $OUT/pw.log 2>&1
cd $REPO
python - "$OUT" > $OUT/summary.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
root = sys.argv[1]
print(open(os.path.join(root, "unprofiled.txt")).read().strip())
vals = {}
for db in sorted(glob.glob(os.path.join(root, "p*", "**", "*.db"), recursive=True)):
    d = sqlite3.connect(db)
    for name, ctr, v, n in d.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                     "where kernel_name like 'bsg::k_build%' group by kernel_name, counter_name"):
        vals.setdefault(name.split("(")[0].replace("bsg::", ""), {})[ctr] = (v, n)
for k, c in sorted(vals.items()):
    print("== %s (%d dispatches)" % (k, max(n for _, n in c.values())))
    waves = c.get("SQ_WAVES", (0, 0))[0]
    for ctr, (v, n) in sorted(c.items()):
        print("   %-24s %16.0f%s" % (ctr, v, ("   per wave %10.1f" % (v / waves)) if waves and ctr != "SQ_WAVES" else ""))
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
        # SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs; SQ_BUSY_CYCLES is summed over the SQs (one per XCD-SE group)
        print("   VALU-active quad-cycles / wave-cycles = %.3f   (ONE wave's view: a SIMD interleaves its resident waves)" % (c["SQ_ACTIVE_INST_VALU"][0] / max(c.get("SQ_WAVE_CYCLES", (1, 0))[0], 1)))
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # the SIMD's view: a wave64 VALU instruction holds its SIMD for 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1 024 SIMDs
        cyc = c["GRBM_GUI_ACTIVE"][0] / 8.0
        print("   VALU issue utilisation of the SIMDs = SQ_INSTS_VALU x 4 / (1 024 x %.0f kernel cycles) = %.3f" % (cyc, c["SQ_INSTS_VALU"][0] * 4 / (1024 * cyc)))
PY
cat $OUT/summary.txt
