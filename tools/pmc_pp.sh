set -u
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
IC=${1:-uniform}; GEO=${2:-cfg3}
rm -rf $R/gpurun_out/pp_pmc
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pp_pmc/$tag -- python3 $R/tests/ppbench.py $IC 3 $GEO > $R/gpurun_out/pp_pmc_$tag.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_pmc/stats -- python3 $R/tests/ppbench.py $IC 3 $GEO > $R/gpurun_out/pp_pmc_stats.log 2>&1
