#!/bin/bash
# SQ counters of the PP kernels on one 560 tile at uniform density (tests/ppbench.py uniform 3 big): tools/pp_pmc.sh
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT; O=/tmp/pppmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  cd $R; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 tests/ppbench.py ${1:-uniform} 3 big steady > $O/$tag.log 2>&1
done
python3 $R/tools/pmc_table.py $O k_pp_light k_pp_ext3 k_pp_intra
