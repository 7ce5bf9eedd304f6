#!/bin/bash
# kernel breakdown of the literal 1024^3 slab transform (bench.py --config slab1024) for library builds: tools/slab_prof.sh base tag ...
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  lib=$R/cubep3m_amd/libp3m_hip_$tag.so; [ "$tag" == base ] && lib=$R/cubep3m_amd/libp3m_hip.so
  rm -rf /tmp/slp; cd $R
  P3M_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/slp -- python3 bench.py --config slab1024 --steps 3 --warmup 1 --no-cpu > /tmp/slp.log 2>&1
  f=$(ls /tmp/slp/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$tag" "$(grep -o '"ms_per_step": [0-9.]*' /tmp/slp.log | head -1)" "${KF_TOP:-12}" <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[4])]
print("%-8s %s" % (sys.argv[2], sys.argv[3]), flush=True)
for r in rows:
    print("    %-70s x%-4s avg %8.1f us  %5s %%" % (r["Name"].split("(")[0].replace("void ", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]), flush=True)
P
done
