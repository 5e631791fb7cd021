#!/bin/bash
# Runs on the GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the ingest / match kernels.
# Usage: tools/profile_ingest_traffic.sh <tag> [n_blocks]
set -u
TAG=$1; NB=${2:-40}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o ing -- python $REPO/tools/ingest_prof.py $NB 10000 1 1 > $OUT/$C.log 2>&1
done
cd $REPO
python - <<PY
import glob, sqlite3
rows = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for d in glob.glob("$OUT/%s/**/*.db" % ctr, recursive=True):
        c = sqlite3.connect(d)
        for k, v, n in c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? and kernel_name like 'bsg::%' group by kernel_name", (ctr,)):
            rows.setdefault(k.split("(")[0].replace("bsg::", ""), {})[ctr] = (v, n)
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/ingest_prof.py $NB 10000 1 1; counter unit KB, per dispatch")
print("# hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950, MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced stream;")
print("# for the scattered 8..32-byte accesses of the table kernels the raw value is the better estimate: both are listed)")
for k, v in sorted(rows.items()):
    f = v.get("FETCH_SIZE", (0, 1)); w = v.get("WRITE_SIZE", (0, 1))
    fk, wk = f[0] / f[1], w[0] / w[1]
    print("%-22s per dispatch: fetch %12.1f KB  write %12.1f KB | all %3d dispatches of the run: raw %9.2f MB  corrected %9.2f MB"
          % (k, fk, wk, f[1], (f[0] + w[0]) * 1024 / 1e6, (2 * f[0] + w[0]) * 1024 / 1e6))
PY
