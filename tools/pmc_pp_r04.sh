#!/bin/bash
# counters of k_pp_ext3 on the clustered and dense inputs, default build and the build that skips the sweep of the heavy records
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_pp_counters.txt; : > $OUT
for ic in ${1:-clustered}; do
  for tag in base nosweep; do
    lib=$R/cubep3m_amd/libp3m_hip.so; [ "$tag" != base ] && lib=$R/cubep3m_amd/libp3m_hip_$tag.so
    export P3M_HIP_LIB=$lib
    bash $R/tools/pmc_pp.sh $ic cfg3 > /dev/null 2>&1
    ( echo "== k_pp_ext3, $ic, cfg3 geometry, build: $tag, per launch"; python3 $R/tools/pmc_table.py $R/gpurun_out/pp_pmc k_pp_ext3; echo ) >> $OUT
  done
done
rm -rf $R/gpurun_out/pp_pmc $R/gpurun_out/pp_pmc_*.log
cat $OUT
