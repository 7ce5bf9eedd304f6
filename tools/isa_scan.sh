#!/bin/bash
# What the compiler made of the memory operations, per kernel: tools/isa_scan.sh [source.hip ...]   (default: every source of the library)
# Compiles each source to gfx950 assembly with make's own flags (device only, into /tmp/p3m_isa) and lists per kernel: vector-memory loads,
# loads that are followed by `s_waitcnt vmcnt(0)` before the next load is issued (independent requests the compiler serialised: a load
# inside a conditional block whose result is merged with a default waits inside the block), flat_* operations (a per-lane choice between
# an LDS and a global address compiled into ONE generic access), ds_bpermute (__shfl_*: an LDS round trip each), scratch bytes.
set -e
cd "$(dirname "$0")/../cubep3m_amd/csrc"
O=/tmp/p3m_isa; mkdir -p $O
srcs="$@"; [ -n "$srcs" ] || srcs=$(make -s -pn | sed -n 's/^SRCS *= *//p' | head -1)
for src in $srcs; do
  b=${src%.hip}
  line=$(make -n -W $src _obj/$b.o | grep -- "-c $src" | head -1)
  [ -n "$line" ] || { echo "no compile line for $src"; continue; }
  eval "${line/-c $src -o _obj\/$b.o/-S --cuda-device-only -o $O/$b.s $src}" 2>/dev/null
done
python3 - $O $srcs <<'PY'
import re, sys
root = sys.argv[1]
print("%-14s %-72s %6s %6s %5s %6s %7s" % ("source", "kernel", "loads", "tight", "flat", "bperm", "scratch"))
for src in sys.argv[2:]:
    b = src[:-4]; lines = open("%s/%s.s" % (root, b)).read().split("\n")
    name = None; st = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m: name = m.group(1); st[name] = [0, 0, 0, 0, 0]; continue
        if name is None: continue
        if re.search(r"\t(global_load|buffer_load|flat_load)", l):
            st[name][0] += 1
            for j in range(i + 1, min(i + 5, len(lines))):
                if re.search(r"\t(global_load|buffer_load|flat_load)", lines[j]): break
                if "s_waitcnt vmcnt(0)" in lines[j]: st[name][1] += 1; break
        if re.search(r"\tflat_", l): st[name][2] += 1
        if "\tds_bpermute" in l: st[name][3] += 1
        m = re.match(r"^; ScratchSize: (\d+)", l)
        if m: st[name][4] = int(m.group(1))
    for k, v in st.items():
        if v[1] >= 3 or v[2] or v[3] or v[4]: print("%-14s %-72s %6d %6d %5d %6d %7d" % (b, k[:72], *v))
PY
