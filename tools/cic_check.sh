#!/bin/bash
# the CIC fine mesh: parity tests that reach it, then the step at the headline's size with its kernel table
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_projection.py tests/test_gpu_group.py -x -q -m gpu -k "cic or CIC or fine_deposit or projection or (config1_kick_parity and pm_cic) or other_tilings or disp_mesh" 2>&1 | tail -4
tools/step_trace.sh cfg4_cic | head -12
