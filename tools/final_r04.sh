#!/bin/bash
# end-of-round run: the whole GPU test suite, the default bench line (all legs, CPU baseline), the round's profiles
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "gpu tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
python3 bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-200 $O/bench_cfg4.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
tools/profile_r04.sh > $O/profile.log 2>&1; tail -n 2 $O/profile.log
