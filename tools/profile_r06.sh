#!/bin/bash
# profiles of round 6 (run on the GPU box through gpurun; outputs under gpurun_out/r06p)
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4pp -- python3 $R/bench.py --config cfg4_pp --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4pp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4cic -- python3 $R/bench.py --config cfg4_cic --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4cic.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4pp_clustered -- python3 $R/tests/stepbench.py cfg4_pp 5 clustered > $O/stats_cfg4pp_clustered.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_write.log 2>&1
# SQ counters of the fused inverse-x + kick pass and of the passes around it (one rocprofv3 pass per counter set)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq/$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra > $O/sq_$tag.log 2>&1
done
python3 $R/tools/pmc_table.py $O/sq k_fft_x_inv2_kick k_fft_lines3r "k_fft_x_inv2<20" k_kick_fix k_zero_many k_scan_lookback > $O/kick_fused_counters.txt 2>&1 || true
cd $R
for c in cfg2 cfg3 cfg2_cic; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; done
python3 bench.py --config slab1024 --steps 5 --warmup 1 > $O/bench_slab1024.json 2> $O/bench_slab1024.err
python3 - <<PY
import csv, glob, statistics, collections
for tag in ("cfg4", "cfg4pp", "cfg4cic", "cfg4pp_clustered"):
    tr = glob.glob("$O/stats_%s/*/*kernel_trace.csv" % tag)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open("$O/%s_kernel_durations.csv" % tag, "w") as f:
        f.write("Kernel_Name,launches,avg_us,median_us\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.2f,%.2f\n' % (k, len(v), sum(v) / len(v), statistics.median(v)))
PY
for t in cfg4 cfg4pp cfg4cic cfg4pp_clustered; do cp $(ls $O/stats_$t/*/*kernel_stats.csv | head -1) $O/${t}_kernel_stats.csv; done
python3 profiles/make_traffic.py $O/pmc cfg4 $O/cfg4 > $O/traffic.log 2>&1 || true
python3 profiles/make_bandwidth.py $O/cfg4_pmc_hbm.csv $O/cfg4_kernel_durations.csv $O/cfg4_bandwidth.md "cfg4 (default bench workload), round 6: achieved HBM bandwidth per kernel" > /dev/null 2>&1 || true
cp profiles/pmc_traffic.json $O/pmc_traffic.json
for d in stats_cfg4 stats_cfg4pp stats_cfg4cic stats_cfg4pp_clustered pmc_FETCH_SIZE pmc_WRITE_SIZE; do rm -f $O/$d/*/*kernel_trace.csv; done
rm -rf $O/sq/*/*/*kernel_trace.csv
: > $O/pp_counters.txt
for ic in uniform clustered dense; do python3 tests/ppbench.py $ic 5 cfg3; done >> $O/pp_counters.txt 2>&1
for ic in uniform clustered; do python3 tests/ppbench.py $ic 5 big steady; done >> $O/pp_counters.txt 2>&1
# SQ counters of the PP kernels (one 560 tile, steady-state order), uniform and clustered; LDS counters of the light pass
bash tools/pp_pmc.sh uniform > $O/pp_sq_uniform.txt 2>&1 || true
bash tools/pp_pmc.sh clustered > $O/pp_sq_clustered.txt 2>&1 || true
bash tools/pp_pmc2.sh uniform > $O/pp_lds_uniform.txt 2>&1 || true
du -sh $O
