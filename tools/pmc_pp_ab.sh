# counters of the extended-PP kernels at uniform density on a 560 tile: k_pp_ext3 (default) and k_pp_ext2 (P3M_PP_EXT_V2=1)
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_pp.sh uniform big > /dev/null 2>&1
( echo "== k_pp_ext3<2> (default), uniform density, nf_tile = 560, per launch"; python3 $R/tools/pmc_table.py $R/gpurun_out/pp_pmc k_pp_ext3 ) > $R/gpurun_out/r03_pp_counters.txt
( echo; echo "== k_pp_intra, same input"; python3 $R/tools/pmc_table.py $R/gpurun_out/pp_pmc k_pp_intra ) >> $R/gpurun_out/r03_pp_counters.txt
export P3M_PP_EXT_V2=1
bash $R/tools/pmc_pp.sh uniform big > /dev/null 2>&1
( echo; echo "== k_pp_ext2<0> (P3M_PP_EXT_V2=1), same input"; python3 $R/tools/pmc_table.py $R/gpurun_out/pp_pmc k_pp_ext2 ) >> $R/gpurun_out/r03_pp_counters.txt
unset P3M_PP_EXT_V2
cd $R
for v in 0 1; do for ic in uniform clustered dense; do echo "P3M_PP_EXT_V2=$v $(P3M_PP_EXT_V2=$v python3 tests/ppbench.py $ic 5 big 2>&1 | tail -1)"; done; done >> $R/gpurun_out/r03_pp_counters.txt
rm -rf $R/gpurun_out/pp_pmc
cat $R/gpurun_out/r03_pp_counters.txt
