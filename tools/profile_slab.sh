# kernel trace of the literal 1024^3 slab-FFT leg (run on the GPU box through gpurun; outputs under gpurun_out/$1)
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-slabp}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --config slab1024 --steps 3 --warmup 1 > $O/stats.log 2>&1
f=$(ls $O/stats/*/*kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; head -40 $O/kernel_stats.csv | cut -c1-200
