"""Per-kernel averages of the counters collected by tools/pmc_step.sh: python tools/pmc_table.py gpurun_out/step_pmc [kernel substring ...]"""
import collections, csv, glob, sys
root = sys.argv[1]; want = sys.argv[2:]
tab = collections.defaultdict(dict)
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if want and not any(w in k for w in want):
            continue
        a = acc[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for (k, c), (n, v) in acc.items():
        tab[k][c] = v / n
for k, d in sorted(tab.items()):
    print(k)
    for c in sorted(d):
        print("   %-34s %16.0f" % (c, d[c]))
