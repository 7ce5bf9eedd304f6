"""Per-step wall time of a bench workload from a cold start: python tests/stepbench.py [config] [nsteps] [uniform|clustered]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cubep3m_amd.params import Params
from cubep3m_amd.group import ParticleMeshGroup
import bench
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "cfg4"
nsteps = int(next((a for a in sys.argv[1:] if a.isdigit()), "8"))
ic = sys.argv[3] if len(sys.argv) > 3 else "uniform"
cfg = bench.CONFIGS[name]; p = Params(**cfg["params"])
grp = ParticleMeshGroup(p, 0, 1)
nside, box = cfg["nside_rank"], float(p.nf_physical_node_dim)
for i, r in enumerate(grp.local_ranks):
    xv = bench.make_particles(nside, box, seed=12345 + r) if ic == "uniform" else bench.clustered(nside, box, 2024 + r, 0.3, 48 * (nside // 32) ** 3, 0.6)
    grp.upload_particles(i, xv, np.arange(1, len(xv) + 1, dtype=np.int64) + r * len(xv))
ts = []
for s in range(nsteps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    grp.particle_mesh(0.5, 0.05, 0.05, 8.0)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("%s %s: steps (ms) %s   median of the last half %.2f" % (name, ic, " ".join("%.1f" % t for t in ts), float(np.median(ts[len(ts) // 2:]))), flush=True)
grp.close()
