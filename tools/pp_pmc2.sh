#!/bin/bash
# LDS counters of the PP kernels on one 560 tile (tests/ppbench.py <ic> 3 big): tools/pp_pmc2.sh [ic]
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT; O=/tmp/pppmc2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  cd $R; rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 tests/ppbench.py ${1:-uniform} 3 big steady > $O/$tag.log 2>&1
  tail -3 $O/$tag.log
done
python3 $R/tools/pmc_table.py $O k_pp_light k_pp_ext3
