#!/bin/bash
# SQ counters of the CIC path's kernels in the step (tests/stepbench.py cfg4_cic 3): tools/cic_pmc.sh [kernel substrings ...]
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT; O=/tmp/cicpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  cd $R; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 tests/stepbench.py ${KF_CFG:-cfg4_cic} 3 uniform > $O/$tag.log 2>&1
done
python3 $R/tools/pmc_table.py $O "$@"
