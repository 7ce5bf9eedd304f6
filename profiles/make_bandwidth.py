"""Per-kernel achieved HBM bandwidth table: PMC bytes per launch (make_traffic.py's *_pmc_hbm.csv) over the launch durations of a
rocprofv3 --kernel-trace run of the same command.
usage: python profiles/make_bandwidth.py <pmc_hbm.csv> <kernel_trace.csv | kernel_durations.csv> <out.md> <title>"""
import collections
import csv
import statistics
import sys


def main(pmc, trace, out, title):
    byt = {r["kernel"]: (int(r["launches"]), float(r["hbm_bytes_per_launch"])) for r in csv.DictReader(open(pmc))}
    dur = collections.defaultdict(list)
    rows = []
    first = next(csv.DictReader(open(trace)))
    if "median_us" in first:                        # a *_kernel_durations.csv (launches, avg, median per kernel) instead of the raw trace
        for r in csv.DictReader(open(trace)):
            if r["Kernel_Name"] in byt:
                rows.append((r["Kernel_Name"], int(r["launches"]), float(r["avg_us"]), float(r["median_us"]), byt[r["Kernel_Name"]][1]))
    else:
        for r in csv.DictReader(open(trace)):
            dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            if k in byt:
                rows.append((k, len(v), sum(v) / len(v), statistics.median(v), byt[k][1]))
    rows.sort(key=lambda r: -r[1] * r[2])
    with open(out, "w") as f:
        f.write("# %s\n\n" % title)
        f.write("Launch durations from the rocprofv3 `--kernel-trace` run named in profiles/README.md (average over all launches incl. the first, unsorted\n"
                "step; median = steady state), bytes per launch from the two PMC passes (`(2*FETCH_SIZE + WRITE_SIZE)*1024`).  TB/s = bytes / median.\n"
                "Kernels of the coarse mesh run on a second stream underneath the fine-mesh passes: their averages include waiting for compute units.\n"
                "Peak 8 TB/s; a non-temporal float4 copy reaches 5.66 TB/s on this part (r02_mallbench.txt).\n\n")
        f.write("| kernel | launches | avg us | median us | GB / launch | TB/s (median) | of 8 TB/s |\n|---|---|---|---|---|---|---|\n")
        for k, n, a, m, b in rows[:24]:
            f.write("| `%s` | %d | %.0f | %.0f | %.3f | %.2f | %.0f %% |\n" % (k.replace("void ", ""), n, a, m, b / 1e9, b / (m * 1e-6) / 1e12, 100 * b / (m * 1e-6) / 8e12))
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:5])
