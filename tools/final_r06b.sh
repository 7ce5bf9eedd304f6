#!/bin/bash
# last run of round 6 on the final code: the whole GPU suite (cold oracle cache), the default bench line (all legs, with roofline.traffic from
# the regenerated profiles/pmc_traffic.json), smoke  (the rocprofv3 passes of the round are tools/profile_r06.sh, run from tools/final_r06.sh)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
export P3M_ORACLE_CACHE=/tmp/oc_final_$$
timeout 1500 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/tests.log 2>&1; echo "gpu tests rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -n 3
python3 bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-200 $O/bench_cfg4.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
