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
    cols = [r[1] for r in d.execute("pragma table_info(kernels)")]
    if all(c in cols for c in ("grid_x", "grid_y", "grid_z")):
        print("== kernel trace durations by launch shape (grid in work-items: x = blocks x 512, z = arenas of the dispatch group)")
        for n, gx, gy, gz, cnt, mean, mn, mx in d.execute(
                "select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration), max(duration) from kernels "
                "where name like 'bsg::k_probe%' or name like 'bsg::k_eval%' group by name, grid_x, grid_y, grid_z order by name, count(*) desc"):
            v = sorted(r[0] for r in d.execute("select duration from kernels where name = ? and grid_x = ? and grid_y = ? and grid_z = ?", (n, gx, gy, gz)))
            print("  %-20s grid (%8d,%4d,%3d)  n %4d  mean %10.1f ns  median %9d  min %9d  max %9d" % (short(n), gx, gy, gz, cnt, mean, v[len(v) // 2], mn, mx))
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

# machine-readable traffic record for bench.py's roofline.traffic (gfx950: FETCH_SIZE of wide coalesced reads x2).
# A probe dispatch covers grid_size_z arenas: every dispatch's counter is divided by its own arena count, so launches of
# different group sizes contribute on an equal footing.
import json
rec = {}
for sub, ctr, key in (("fetch", "FETCH_SIZE", "fetch_kb_raw"), ("write", "WRITE_SIZE", "write_kb_raw")):
    d = db(sub)
    if not d:
        continue
    for n in [r[0] for r in d.execute("select distinct kernel_name from counters_collection where kernel_name like 'bsg::%'")]:
        rows = list(d.execute("select value, grid_size_z from counters_collection where kernel_name = ? and counter_name = ? order by start desc limit 200", (n, ctr)))
        if not rows:
            continue
        e = rec.setdefault(short(n), {})
        e[key] = sum(v for v, _ in rows) / len(rows)
        e[key + "_per_arena"] = sum(v / max(z, 1) for v, z in rows) / len(rows)
        e["dispatches"] = len(rows)
        e["arenas_per_dispatch_max"] = max(z for _, z in rows)
for k, v in rec.items():
    v["hbm_bytes_corrected"] = (2 * v.get("fetch_kb_raw", 0) + v.get("write_kb_raw", 0)) * 1024
    v["hbm_bytes_corrected_per_arena"] = (2 * v.get("fetch_kb_raw_per_arena", 0) + v.get("write_kb_raw_per_arena", 0)) * 1024
json.dump(rec, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print("== traffic.json:", json.dumps(rec))
