#!/bin/bash
# A/B of an environment switch on the five passes and the step: ab_env.sh VAR
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for v in 0 1; do
    env $1=$v python3 bench.py --no-cpu --no-extra --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$1=$v]', round(d['ms_per_step'],3), {k: round(v,4) for k,v in r.get('pass_ms',{}).items()})"
  done
done
