"""GPU-busy fraction of the timed steps from a rocprofv3 --kernel-trace csv: python tools/busy.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# last <steps> occurrences of the first kernel of a step mark the step starts
steps = int(sys.argv[2])
name0 = "k_compact_drift_hist"
starts = [s for s, e, n in rows if n.startswith(name0)]
if len(starts) < steps + 1:
    print("not enough steps", len(starts)); sys.exit()
t0, t1 = starts[-steps - 1], starts[-1]
sel = [(s, e) for s, e, n in rows if s >= t0 and s < t1]
# union of intervals (two streams overlap)
busy = 0; cur_s, cur_e = None, None
for s, e in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("steps %d  wall %.3f ms/step  busy %.3f ms/step  (%.1f %%)  launches/step %.0f" % (steps, (t1 - t0) / steps / 1e6, busy / steps / 1e6, 100.0 * busy / (t1 - t0), len(sel) / steps))
