#!/bin/bash
# diagnostic: do kernels of ONE queue ever overlap in time?  kernel trace of tests/determinism_check.py, then per queue the launches
# whose start precedes the end of the launch before them
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ovl
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -- python3 tests/determinism_check.py 4 cfg4 4 > /tmp/ovl.log 2>&1
grep "rep" /tmp/ovl.log | cut -c1-120
python3 - <<'P'
import csv, glob, collections
f = glob.glob("/tmp/ovl/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]))
for q, v in byq.items():
    v.sort()
    n = 0
    for (s0, e0, k0), (s1, e1, k1) in zip(v, v[1:]):
        if s1 < e0:
            n += 1
            if n <= 12: print("queue %s: %s [%d..%d] overlaps the next launch %s [%d..%d] by %d ns" % (q, k0, s0, e0, k1, s1, e1, e0 - s1))
    print("queue %s: %d launches, %d overlapping pairs" % (q, len(v), n))
P
