#!/bin/bash
# coarse slab transform: its tests, the bench line and a kernel trace of it -> gpurun_out/slab
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/slab; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_slab1024.py -x -q -m gpu > $O/tests.log 2>&1; echo "slab tests rc=$?" | tee -a $O/tests.log
if [ "${1:-}" = group ]; then timeout 1500 python3 -m pytest tests/test_gpu_group.py -x -q -m gpu > $O/group.log 2>&1; echo "group tests rc=$?" | tee -a $O/group.log; fi
python3 bench.py --config slab1024 --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; cut -c1-220 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --config slab1024 --steps 5 --warmup 1 > $O/stats.log 2>&1
rm -f $O/stats/*/*kernel_trace.csv
python3 - <<PY
import csv, glob
for r in list(csv.DictReader(open(glob.glob("$O/stats/*/*kernel_stats.csv")[0])))[:16]:
    print("%-58s %4s %9.3f ms avg %8.1f us" % (r["Name"].split("(")[0][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
