# profiles of round 3 (run on the GPU box through gpurun; outputs under gpurun_out/r03p)
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4pp -- python3 $R/bench.py --config cfg4_pp --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4pp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4cic -- python3 $R/bench.py --config cfg4_cic --steps 4 --warmup 2 --no-cpu --no-extra > $O/stats_cfg4cic.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_slab1024 -- python3 $R/bench.py --config slab1024 --steps 3 --warmup 1 > $O/stats_slab1024.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmccic_FETCH_SIZE -- python3 $R/bench.py --config cfg4_cic --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmccic_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmccic_WRITE_SIZE -- python3 $R/bench.py --config cfg4_cic --steps 2 --warmup 2 --no-cpu --no-extra > $O/pmccic_write.log 2>&1
cd $R
for c in cfg2 cfg3 cfg2_cic; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; done
python3 bench.py --config slab1024 --steps 5 --warmup 1 > $O/bench_slab1024.json 2> $O/bench_slab1024.err
python3 bench.py --steps 10 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
# keep the merge small: medians instead of the raw traces (the cfg4 trace is kept for the bandwidth table)
for d in stats_cfg4pp stats_cfg4cic stats_slab1024 pmc_FETCH_SIZE pmc_WRITE_SIZE pmccic_FETCH_SIZE pmccic_WRITE_SIZE; do rm -f $O/$d/*/*kernel_trace.csv; done
python3 - <<PY
import csv, glob, statistics, collections
tr = glob.glob("$O/stats_cfg4/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$O/cfg4_kernel_durations.csv", "w") as f:   # compact stand-in for the 100 MB trace: name, launches, average, median
    f.write("Kernel_Name,launches,avg_us,median_us\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%.2f,%.2f\n' % (k, len(v), sum(v) / len(v), statistics.median(v)))
PY
ls -la $O/stats_cfg4/*/ | head; du -sh $O
