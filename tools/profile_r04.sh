#!/bin/bash
# profiles of round 4 (run on the GPU box through gpurun; outputs under gpurun_out/r04p)
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4pp -- python3 $R/bench.py --config cfg4_pp --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4pp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4cic -- python3 $R/bench.py --config cfg4_cic --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4cic.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4pp_clustered -- python3 $R/tests/stepbench.py cfg4_pp 5 clustered > $O/stats_cfg4pp_clustered.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmccic_FETCH_SIZE -- python3 $R/bench.py --config cfg4_cic --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmccic_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmccic_WRITE_SIZE -- python3 $R/bench.py --config cfg4_cic --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmccic_write.log 2>&1
cd $R
for c in cfg2 cfg3 cfg2_cic; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; done
python3 bench.py --config slab1024 --steps 5 --warmup 1 > $O/bench_slab1024.json 2> $O/bench_slab1024.err
# keep the merge small: medians instead of the raw traces
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
for d in stats_cfg4 stats_cfg4pp stats_cfg4cic stats_cfg4pp_clustered pmc_FETCH_SIZE pmc_WRITE_SIZE pmccic_FETCH_SIZE pmccic_WRITE_SIZE; do rm -f $O/$d/*/*kernel_trace.csv; done
# counters of the two passes of the extended PP (pass 0: lists and walks; pass 1: sweeps) on the three inputs of the pp leg
: > $O/pp_counters.txt
for ic in uniform clustered dense; do
  bash $R/tools/pmc_pp.sh $ic cfg3 > /dev/null 2>&1 || true
  ( echo "== k_pp_ext3 (pass 0 = <2, true, 0>, pass 1 = <2, true, 1>), $ic, cfg3 geometry, per launch"; python3 $R/tools/pmc_table.py $R/gpurun_out/pp_pmc k_pp_ext3; echo ) >> $O/pp_counters.txt
done
for ic in uniform clustered dense; do python3 tests/ppbench.py $ic 5 cfg3; done >> $O/pp_counters.txt 2>&1
for ic in uniform clustered; do python3 tests/ppbench.py $ic 3 big; done >> $O/pp_counters.txt 2>&1
rm -rf $R/gpurun_out/pp_pmc $R/gpurun_out/pp_pmc_*.log
du -sh $O
