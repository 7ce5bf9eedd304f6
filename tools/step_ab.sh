# kernel averages of the default step under rocprofv3: bash tools/step_ab.sh <tag> <kernel substrings...>
# (P3M_HIP_LIB, if set, must be an absolute path: the run starts in /tmp)
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$tag -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu --no-extra > $R/gpurun_out/ab_$tag.log 2>&1
f=$(find $R/gpurun_out/ab_$tag -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "$tag: the run failed:"; tail -3 $R/gpurun_out/ab_$tag.log; exit 1; fi   # (an empty $f would leave grep reading stdin)
echo "$tag $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/ab_$tag.log)"
for k in "$@"; do grep "$k" $f | awk -F, -v k=$k '{printf "   %-28s calls %s avg %.1f us min %.1f us\n", k, $(NF-6), $(NF-4)/1000, $(NF-2)/1000}'; done
