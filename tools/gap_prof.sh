#!/bin/bash
# how much of a step no kernel covers: rocprofv3 kernel trace of tests/stepbench.py, union of all kernel intervals over the last steps
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python3 tests/stepbench.py ${KF_CFG:-cfg4} 6 uniform > /tmp/gp.log 2>&1
grep median /tmp/gp.log
python3 - "$(ls /tmp/gp/*/*kernel_trace.csv | head -1)" <<'P'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the steps: split at the fused kick launches (8 per step); take the span of the last 3 steps
kicks = [i for i, r in enumerate(rows) if "k_fft_x_inv2_kick" in r[2]]
nstep = len(kicks) // 8
first = kicks[(nstep - 3) * 8 - 1] + 1 if nstep > 3 else 0     # after the last kick of step nstep-4 ... approximately a step boundary
seg = rows[first:kicks[-1] + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
gaps = []
for s, e, name, q, st in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, name)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms over ~3 steps, covered by kernels %.2f ms, idle %.2f ms (%.1f %%), %d launches, %d gaps" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / (t1 - t0), len(seg), len(gaps)))
gaps.sort(reverse=True)
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:12]])
hist = collections.Counter(min(int(g / 1e3) // 2 * 2, 40) for g, n in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
by = collections.Counter()
for g, n in gaps: by[n[:48]] += g
print("idle before kernel (us total):", [(n, round(v / 1e3)) for n, v in by.most_common(12)])
P
