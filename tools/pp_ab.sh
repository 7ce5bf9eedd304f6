#!/bin/bash
# A/B timing of library variants built by tools/variant.sh: tools/pp_ab.sh "<tag> <tag> ..." "<ic> <ic> ..." [cfg3|big]
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pp_ab; mkdir -p $O
geo=${3:-cfg3}
for tag in $1; do
  lib=cubep3m_amd/libp3m_hip.so; [ "$tag" != base ] && lib=cubep3m_amd/libp3m_hip_$tag.so
  for ic in $2; do echo -n "[$tag] "; P3M_HIP_LIB=$PWD/$lib timeout 300 python3 tests/ppbench.py $ic 5 $geo ${4:-}; done
done 2>&1 | tee $O/ab_$geo.log
