"""Summarise a tools/profile.sh output directory (rocprofv3 rocpd sqlite output): per-kernel
duration stats from the kernel trace, and per-dispatch FETCH_SIZE / WRITE_SIZE from the PMC passes."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]


def db(sub):
    hits = glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


def short(name):
    return name.split("(")[0].replace("bsg::", "")


d = db("stats")
if d:
    print("== rocprofv3 --kernel-trace --stats : top_kernels (name, calls, total us, avg us, %)")
    for name, calls, total, avg, pct in d.execute("select * from top_kernels"):
        print("  %-28s calls %5d  total %10.1f us  avg %9.3f us  %5.1f%%" % (short(name), calls, total, avg, pct))
    print("== kernel trace durations (end - start, ns), last 200 dispatches of each bsg kernel")
    names = [r[0] for r in d.execute("select distinct name from kernels where name like 'bsg::%'")]
    for n in names:
        v = sorted(r[0] for r in d.execute("select duration from (select duration, start from kernels where name = ? order by start desc limit 200)", (n,)))
        print("  %-28s n %4d  mean %9.1f  median %8d  p10 %8d  p90 %8d  grid %s" % (
            short(n), len(v), sum(v) / len(v), v[len(v) // 2], v[len(v) // 10], v[len(v) * 9 // 10],
            d.execute("select grid_x, grid_y, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels where name = ? order by start desc limit 1", (n,)).fetchone()))
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    d = db(sub)
    if not d:
        print("== no database for", ctr)
        continue
    print("== %s per dispatch (counter unit KB), last 200 dispatches" % ctr)
    names = [r[0] for r in d.execute("select distinct kernel_name from counters_collection where kernel_name like 'bsg::%'")]
    for n in names:
        v = [r[0] for r in d.execute("select value from (select value, start from counters_collection where kernel_name = ? and counter_name = ? order by start desc limit 200)", (n, ctr))]
        if v:
            print("  %-28s n %4d  mean %12.1f KB = %9.3f MB  (min %.1f max %.1f)" % (short(n), len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / 1e6, min(v), max(v)))

# machine-readable traffic record for bench.py's roofline.traffic (gfx950: FETCH_SIZE of wide coalesced reads x2)
import json
rec = {}
d = db("fetch")
if d:
    for n in [r[0] for r in d.execute("select distinct kernel_name from counters_collection where kernel_name like 'bsg::%'")]:
        v = [r[0] for r in d.execute("select value from (select value, start from counters_collection where kernel_name = ? and counter_name = 'FETCH_SIZE' order by start desc limit 200)", (n,))]
        if v:
            rec.setdefault(short(n), {})["fetch_kb_raw"] = sum(v) / len(v)
d = db("write")
if d:
    for n in [r[0] for r in d.execute("select distinct kernel_name from counters_collection where kernel_name like 'bsg::%'")]:
        v = [r[0] for r in d.execute("select value from (select value, start from counters_collection where kernel_name = ? and counter_name = 'WRITE_SIZE' order by start desc limit 200)", (n,))]
        if v:
            rec.setdefault(short(n), {})["write_kb_raw"] = sum(v) / len(v)
for k, v in rec.items():
    v["hbm_bytes_corrected"] = (2 * v.get("fetch_kb_raw", 0) + v.get("write_kb_raw", 0)) * 1024
json.dump(rec, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print("== traffic.json:", json.dumps(rec))
