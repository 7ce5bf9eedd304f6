#!/bin/bash
# instruction-fetch / wait counters of the PP kernels on one 560 tile: tools/pp_pmc3.sh [ic] ["lib env"]
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT; O=/tmp/pppmc3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC" "SQ_INSTS_BRANCH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  cd $R; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 tests/ppbench.py ${1:-uniform} 3 big > $O/$tag.log 2>&1
done
python3 $R/tools/pmc_table.py $O k_pp_light k_pp_ext3
