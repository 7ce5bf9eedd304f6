"""Per-step wall time of the default bench workload from a cold start: python tests/stepbench.py [nsteps]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from cubep3m_amd.params import Params
from cubep3m_amd.group import ParticleMeshGroup
import bench
cfg = bench.CONFIGS["cfg4"]; p = Params(**cfg["params"])
grp = ParticleMeshGroup(p, 0, 1)
for i, r in enumerate(grp.local_ranks):
    xv = bench.make_particles(cfg["nside_rank"], float(p.nf_physical_node_dim), seed=12345 + r)
    grp.upload_particles(i, xv, np.arange(1, len(xv) + 1, dtype=np.int64) + r * len(xv))
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    grp.particle_mesh(0.5, 0.05, 0.05, 8.0)
    torch.cuda.synchronize(); print("step %d: %.2f ms" % (s, 1e3 * (time.perf_counter() - t0)), flush=True)
grp.close()
