#!/bin/bash
# P3M_GROUP_STREAMS=8 (a stream per logical rank) against the default: the step, and what rocprofv3 then reads as the kernels' durations
# (kernels of different ranks share the device: a kernel's duration is no longer its own).  Run through gpurun: bash tools/streams_ab.sh
set -u
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 8 0; do
rm -rf /tmp/st$v; P3M_GROUP_STREAMS=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st$v -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > /tmp/st$v.log 2>&1
echo "== P3M_GROUP_STREAMS=$v  $(tail -1 /tmp/st$v.log | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
python3 - /tmp/st$v <<'P'
import csv, glob, sys, statistics, collections
tr = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)): d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]: print("   %-55s n %3d avg %8.1f med %8.1f" % (k[:55], len(v), sum(v)/len(v), statistics.median(v)))
P
done
