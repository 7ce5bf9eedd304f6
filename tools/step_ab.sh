#!/bin/bash
# whole-step A/B of library builds (tools/variant.sh): per-kernel averages of the hottest kernels and the step's median
#   tools/step_ab.sh base tag1 tag2 ...      KF_CFG (cfg4), KF_IC (uniform), KF_TOP (14 kernels)
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rnd in $(seq 1 ${KF_ROUNDS:-2}); do
for tag in "$@"; do
  lib=$R/cubep3m_amd/libp3m_hip_$tag.so; [ "$tag" == base ] && lib=$R/cubep3m_amd/libp3m_hip.so
  rm -rf /tmp/kfab; cd $R
  P3M_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kfab -- python3 tests/stepbench.py ${KF_CFG:-cfg4} 4 ${KF_IC:-uniform} > /tmp/kfab.log 2>&1
  f=$(ls /tmp/kfab/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$rnd" "$tag" "$(grep median /tmp/kfab.log | sed 's/.*median/median/')" "${KF_TOP:-14}" <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[5])]
print("round %s %-8s %s" % (sys.argv[2], sys.argv[3], sys.argv[4]), flush=True)
for r in rows:
    print("    %-62s x%-4s avg %8.1f us" % (r["Name"].split("(")[0].replace("void ", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3), flush=True)
P
done
done
