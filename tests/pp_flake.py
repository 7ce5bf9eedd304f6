"""Repeats one whole step of the extended-PP cases from the same input and reports what differs between the runs:
python tests/pp_flake.py [pp_range] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from common import cfg1, clustered_particles
from cubep3m_amd.particle_mesh import ParticleMesh
ppr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = cfg1(tiles_node_dim=2, nf_tile=80, cores=2, ngp=True, ppint=True, pp_ext=True, pp_range=ppr)
box = float(p.nf_physical_node_dim); n = int(box ** 3 / 8)
xv = clustered_particles(n, box, seed=102, frac=0.3, nblobs=20, sigma=1.0, vel_sigma=0.5)
pid = np.arange(1, n + 1, dtype=np.int64)
ref = None
for r in range(reps):
    g = ParticleMesh(p)
    g.upload_particles(xv, pid)
    outs = [g.particle_mesh(0.2, 0.05, 0.04 if s == 0 else 0.05, 8.0) for s in range(2)]
    x, q = g.download_particles()
    o = np.argsort(q); x = x[o]
    key = (outs[0].dt_pp_ext_acc, outs[1].dt_pp_ext_acc)
    if ref is None: ref = x
    d = np.abs(x[:, 3:] - ref[:, 3:]).max(axis=1)
    bad = np.nonzero(d > 0)[0]
    print("run %2d dt_pp_ext_acc %.6g %.6g  dt_pp %.6g  particles differing from run 0: %d  max |dv| %.3g %s" % (
        r, key[0], key[1], outs[1].dt_pp_acc, len(bad), d.max(), ("ids " + str(q[o][bad][:8])) if len(bad) else ""), flush=True)
    g.close()
