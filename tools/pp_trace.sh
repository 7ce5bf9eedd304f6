#!/bin/bash
# kernel trace of tests/ppbench.py: tools/pp_trace.sh <ic> <geo>  ->  gpurun_out/pp_trace_<ic>_<geo>.csv (per-kernel statistics)
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ic=${1:-clustered}; geo=${2:-big}
rm -rf /tmp/pp_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_trace -- python3 $R/tests/ppbench.py $ic 3 $geo > /tmp/pp_trace.log 2>&1
f=$(ls /tmp/pp_trace/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/pp_trace_${ic}_${geo}.csv
head -12 $f | cut -c1-60,200-400
