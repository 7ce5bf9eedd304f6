"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a bench.py run into per-kernel HBM bytes per launch.

usage: python profiles/make_traffic.py <dir with *_FETCH_SIZE/ and *_WRITE_SIZE/ rocprofv3 outputs prefix> <config> <out prefix>
  e.g. python profiles/make_traffic.py gpurun_out/i_pmc cfg4 profiles/r01_e_cfg4

bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as
64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the factor 2 on the read side.
Writes <out prefix>_pmc_hbm.csv and updates profiles/pmc_traffic.json[config] (read by bench.py for roofline.traffic).
"""
import csv, glob, hashlib, json, os, sys


def fft_source_sha16():
    """Identity of the FFT pass kernels the byte counts belong to: bench.py reports roofline.traffic only while the
    sources it runs hash to the same value (anything else is a stale file and reads as null)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("fft.hip", "fft_core.h", "fft_x2.h", "kick_fused.hip"):
        h.update(open(os.path.join(root, "cubep3m_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]

PASS_OF = [  # kernel-name prefix (template arguments included where they tell passes apart) -> FFT pass of the fine sweep
    ("void k_fft_x_fwd", "x_fwd"), ("void k_fft_x_inv2_kick", "x_inv_kick_fused"), ("void k_fft_x_inv", "x_inv_extract"), ("void k_fft_lines3", "z_inv_fused"),
]


def pass_of(name):
    for pre, ps in PASS_OF:
        if name.startswith(pre):
            return ps
    if name.startswith("void k_fft_lines2<"):       # <R1, R2, INV, TR>
        a = [x.strip() for x in name[name.index("<") + 1:name.index(">")].split(",")]
        return {("false", "true"): "y_fwd", ("true", "false"): "y_inv", ("false", "false"): "z_fwd", ("true", "true"): "z_inv"}[(a[2], a[3])]
    if name.startswith("void k_fft_lines<"):        # <INV, TR, NC, ...>
        a = [x.strip() for x in name[name.index("<") + 1:name.index(">")].split(",")]
        return {("false", "true"): "y_fwd", ("true", "false"): "y_inv", ("false", "false"): "z_fwd", ("true", "true"): "z_inv"}[(a[0], a[1])]
    return None


def main(prefix, config, out):
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s_%s/*/*_counter_collection.csv" % (prefix, ctr))[0]
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            a = per.setdefault(k, {}).setdefault(ctr, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    rows = []
    for k, d in per.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        f = d["FETCH_SIZE"][1] / d["FETCH_SIZE"][0]
        w = d["WRITE_SIZE"][1] / d["WRITE_SIZE"][0]
        rows.append((k, d["FETCH_SIZE"][0], f, w, (2 * f + w) * 1024))
    rows.sort(key=lambda r: -r[4] * r[1])
    with open(out + "_pmc_hbm.csv", "w") as fo:
        fo.write("kernel,launches,FETCH_SIZE_avg_raw_KB,WRITE_SIZE_avg_KB,hbm_bytes_per_launch\n")
        for r in rows[:48]:
            fo.write('"%s",%d,%.1f,%.1f,%.0f\n' % r)
    passes = {}
    cand = {}
    for k, nl, f, w, b in rows:
        ps = pass_of(k)
        if ps:
            cand.setdefault(ps, []).append((k, nl, f, w, b))
    for ps, cs in cand.items():
        # the fine-mesh launch is the big one (the coarse mesh shares kernel templates); where a pass exists in two variants of the fine mesh
        # -- round 6: the forward x pass reading floats (the first step after an upload) or one byte per cell (every later step) -- the
        # variant the steps run most
        big = max(c[4] for c in cs)
        k, nl, f, w, b = max((c for c in cs if c[4] >= 0.2 * big), key=lambda c: (c[1], c[4]))
        passes[ps] = {"kernel": k, "launches": nl, "hbm_bytes_per_launch": b, "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024}
    tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    allj = json.load(open(tf)) if os.path.exists(tf) else {}
    allj[config] = {"passes": passes, "fft_source_sha16": fft_source_sha16(), "source": os.path.relpath(out + "_pmc_hbm.csv", os.path.dirname(tf) + "/.."),
                    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 2 --warmup 2 --no-cpu`; "
                              "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 counts a 128-byte read request as 64 bytes)"}
    json.dump(allj, open(tf, "w"), indent=1)
    for ps, d in passes.items():
        print("%-14s %-48s %.3f GB/launch" % (ps, d["kernel"][:48], d["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main(*sys.argv[1:4])
