# profiles of round 2 (run on the GPU box through gpurun; outputs under gpurun_out/r02p)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4pp -- python3 $R/bench.py --config cfg4_pp --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4pp.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_write.log 2>&1
cd $R
for c in cfg2 cfg3 cfg2_cic; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; done
python3 bench.py --steps 10 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
