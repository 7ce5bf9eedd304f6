#!/bin/bash
# kernel time of k_fft_x_inv2_kick for several builds of kick_fused.hip (tools/variant.sh): tools/kf_ab.sh base tag1 tag2 ...
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2; do
for tag in "$@"; do
  lib=$R/cubep3m_amd/libp3m_hip_$tag.so; [ "$tag" == base ] && lib=$R/cubep3m_amd/libp3m_hip.so
  rm -rf /tmp/kfab; cd $R
  P3M_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kfab -- python3 tests/stepbench.py ${KF_CFG:-cfg4} 4 ${KF_IC:-uniform} > /tmp/kfab.log 2>&1
  f=$(ls /tmp/kfab/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$rnd" "$tag" "$(grep median /tmp/kfab.log | sed 's/.*median/median/')" <<'P'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(s in n for s in ("k_fft_x_inv2_kick", "k_fine_kick_rows", "k_fft_x_inv2<", "k_kick_fix")):
        out.append("%s x%s avg %.1f min %.1f us" % (n.split("(")[0].replace("void ", "")[:30], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
print("round %s %-10s %s | %s" % (sys.argv[2], sys.argv[3], sys.argv[4], " | ".join(out)), flush=True)
P
done
done
