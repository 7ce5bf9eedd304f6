# LDS counters of the default step's kernels (unaligned / bank-conflict stalls of the exchange buffers): tools/fft_pmc_lds.sh
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fft_lds
rm -rf $O; mkdir -p $O
for set in "SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra > $O/$tag.log 2>&1 || tail -5 $O/$tag.log
done
python3 $R/tools/pmc_table.py $O k_fft k_row k_coarse k_compact > $O/table.txt
cat $O/table.txt
