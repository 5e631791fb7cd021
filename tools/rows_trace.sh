#!/bin/bash
# kernel-trace of one bench run; prints the durations of k_survivor_rows (and its neighbours) grouped by grid size
set -e
REPO=$(pwd)
export TMPDIR=/tmp
OUT=/tmp/rows_trace
rm -rf $OUT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 20 --warmup 5 --no-big-filters > /tmp/rows_trace.json 2> /tmp/rows_trace.err) || { tail -5 /tmp/rows_trace.err; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/rows_trace/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'k_survivor_rows' in n or 'k_eval_programs' in n:
        key = (n.split('(')[0][-40:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print(k, 'n=%d median %.2f us min %.2f max %.2f' % (len(v), v[len(v)//2], v[0], v[-1]))
PY
