#!/bin/bash
# in-step A/B of library variants (tools/variant.sh): tools/step_ab.sh "<tag> ..." [config] [nsteps] [uniform|clustered] [rounds]
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/step_ab; mkdir -p $O
cfg=${2:-cfg4_pp}; n=${3:-12}; ic=${4:-uniform}; rounds=${5:-2}
for r in $(seq $rounds); do
  for tag in $1; do
    lib=cubep3m_amd/libp3m_hip.so; [ "$tag" != base ] && lib=cubep3m_amd/libp3m_hip_$tag.so
    echo -n "[$tag] "; P3M_HIP_LIB=$PWD/$lib timeout 600 python3 tests/stepbench.py $cfg $n $ic || true
  done
done 2>&1 | tee $O/ab_${cfg}_$ic.log
