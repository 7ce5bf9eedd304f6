#!/bin/bash
# PP kernels: the parity tests that reach them, the stand-alone timings of tests/ppbench.py at three ICs, two geometries.
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pp_check; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "p3m_ext or dense_blob or rim_planes or two_steps_with_drift or other_tilings or config1_kick_parity or half_an_ulp or force_maximum_repeats" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
tail -5 $O/parity.log
for ic in uniform clustered dense; do timeout 300 python3 tests/ppbench.py $ic 5 cfg3; done 2>&1 | tee $O/ppbench_cfg3.log
for ic in uniform clustered; do timeout 300 python3 tests/ppbench.py $ic 3 big; done 2>&1 | tee $O/ppbench_big.log
for ic in uniform clustered; do echo -n "[P3M_PP_LIGHT_OFF] "; P3M_PP_LIGHT_OFF=1 timeout 300 python3 tests/ppbench.py $ic 3 big; done 2>&1 | tee $O/ppbench_big_off.log
