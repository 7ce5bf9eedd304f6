"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> ra.txt; python tools/regusage.py ra.txt [substring ...]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
pats = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', t)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, dn in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    dn = dn.split('(')[0].replace('void ', '')
    if pats and not any(p in dn for p in pats):
        continue
    sc, oc, ld = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')
    print(f"{dn:64s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):3d} scratch={sc:4d} occ={oc} lds={ld}")
