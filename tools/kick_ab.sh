#!/bin/bash
# A/B of the fused inverse-x + kick pass against the force box + k_fine_kick_rows pair: tools/kick_ab.sh cfg1 cfg2 cfg4_small ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "$@"; do
  for ic in uniform clustered; do
    P3M_KICK_UNFUSED=1 python3 tests/kick_ab.py $cfg /tmp/kab_a.npz $ic > gpurun_out/kick_ab_${cfg}_${ic}.log 2>&1
    python3 tests/kick_ab.py $cfg /tmp/kab_b.npz $ic >> gpurun_out/kick_ab_${cfg}_${ic}.log 2>&1
    python3 tests/kick_ab.py --cmp /tmp/kab_a.npz /tmp/kab_b.npz >> gpurun_out/kick_ab_${cfg}_${ic}.log 2>&1
    echo "$cfg $ic rc=$?"; tail -4 gpurun_out/kick_ab_${cfg}_${ic}.log
  done
done
