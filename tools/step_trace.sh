#!/bin/bash
# kernel statistics of tests/stepbench.py: tools/step_trace.sh <config> [uniform|clustered]  ->  gpurun_out/step_trace_<config>_<ic>.csv
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cfg=${1:-cfg4}; ic=${2:-uniform}
rm -rf /tmp/step_trace
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_trace -- python3 tests/stepbench.py $cfg 6 $ic > /tmp/step_trace.log 2>&1
tail -1 /tmp/step_trace.log
f=$(ls /tmp/step_trace/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/step_trace_${cfg}_${ic}.csv
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-60s calls %5s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
P
