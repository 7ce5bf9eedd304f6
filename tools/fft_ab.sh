#!/bin/bash
# A/B of library variants (tools/variant.sh) on the five passes of the fine sweep and the step: tools/fft_ab.sh "<tag> ..." [config]
set -eu
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fft_ab; mkdir -p $O
cfg=${2:-cfg4}
for r in 1 2; do
  for tag in $1; do
    lib=cubep3m_amd/libp3m_hip.so; [ "$tag" != base ] && lib=cubep3m_amd/libp3m_hip_$tag.so
    P3M_HIP_LIB=$PWD/$lib python3 bench.py --config $cfg --no-cpu --no-extra --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$tag]', round(d['ms_per_step'],3), {k: round(v,4) for k,v in r.get('pass_ms',{}).items()}, 'sweep', r.get('fine_sweep',{}).get('ms'))"
  done
done 2>&1 | tee $O/ab_$cfg.log
