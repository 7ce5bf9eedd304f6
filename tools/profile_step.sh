# kernel trace of the default step (run on the GPU box through gpurun; outputs under gpurun_out/$1); $2: extra bench args
# prints calls / average / MEDIAN per kernel (the first step sorts unsorted records: its launches inflate the averages)
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-stepp}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra $2 > $O/stats.log 2>&1
f=$(ls $O/stats/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv
python3 - <<PY
import csv, glob, statistics, collections
tr = glob.glob("$O/stats/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in d.values())
with open("$O/kernel_medians.csv", "w") as f:
    f.write("Name,Calls,TotalUs,AverageUs,MedianUs,Percentage\n")
    for k, v in rows:
        f.write('"%s",%d,%.1f,%.1f,%.1f,%.2f\n' % (k, len(v), sum(v), sum(v) / len(v), statistics.median(v), 100 * sum(v) / tot))
for k, v in rows[:24]:
    print("%-84s %5d avg %8.1f med %8.1f  %5.2f%%" % (k[:84], len(v), sum(v) / len(v), statistics.median(v), 100 * sum(v) / tot))
PY
rm -rf $O/stats/*/*kernel_trace.csv
