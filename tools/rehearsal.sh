#!/bin/bash
# cfg4 at FULL size as 2, 4 and 8 processes sharing the one GPU (gloo host transport: pinned-host staging, the
# message sizes and 64-bit offsets of a real multi-process run), one line each -> gpurun_out/r04_bench_cfg4_{2,4,8}proc_gloo.json
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
for n in 2 4 8; do
  timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 3 --warmup 1 \
      --no-cpu --no-extra --dist-backend gloo > gpurun_out/r04_bench_cfg4_${n}proc_gloo.json 2> gpurun_out/r04_bench_cfg4_${n}proc_gloo.err
  echo "n=$n rc=$? $(head -c 300 gpurun_out/r04_bench_cfg4_${n}proc_gloo.json)"
done
