"""Diagnostic: the same full-size 8-rank cfg4 step from fresh groups several times -- the dt limits must repeat bit for bit.
    python tests/determinism_check.py [reps]      (P3M_ONE_STREAM=1 in the environment: without the second stream)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cubep3m_amd.params import Params
from cubep3m_amd.group import ParticleMeshGroup
import bench
p = Params(**bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg4"]["params"])
nside = 256
box = float(p.nf_physical_node_dim)
parts = [bench.make_particles(nside, box, seed=4000 + r) for r in range(p.nodes)]
vs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
for r, x in enumerate(parts):
    x[:, 3:] = np.random.default_rng(900 + r).normal(0, vs, (len(x), 3)).astype(np.float32)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    g = ParticleMeshGroup(p, 0, 1)
    for i, r in enumerate(g.local_ranks):
        g.upload_particles(i, parts[r], np.arange(1, len(parts[r]) + 1, dtype=np.int64) + r * len(parts[r]))
    o1 = g.particle_mesh(0.5, 0.05, 0.0, 8.0)
    o2 = g.particle_mesh(0.5, 0.05, 0.05, 8.0)
    if o2.dt_c_acc < 2.0:
        for i in range(len(g.local_ranks)):
            fc = g.coarse_force(i).astype(np.float64)
            mag = np.sqrt((fc ** 2).sum(-1))
            k = np.unravel_index(np.argmax(mag), mag.shape)
            big = np.argwhere(mag > 20.0)
            if len(big):
                z0, y0 = int(big[0, 0]), int(big[0, 1])
                for yy in range(y0 - 1, y0 + 5):
                    print("      fc[z=%d, y=%d, x=1..6]: comp0 %s comp1 %s comp2 %s" % (z0, yy, np.round(fc[z0, yy, 1:7, 0], 2), np.round(fc[z0, yy, 1:7, 1], 2), np.round(fc[z0, yy, 1:7, 2], 2)), flush=True)
            print("   rank %d: max |F_c| %.4g at (z,y,x) %s, %d cells above 20; z range %s y range %s x range %s" %
                  (i, mag.max(), k, len(big), (big[:, 0].min(), big[:, 0].max()) if len(big) else None, (big[:, 1].min(), big[:, 1].max()) if len(big) else None,
                   (big[:, 2].min(), big[:, 2].max()) if len(big) else None), flush=True)
    print("rep %d: step 1 dt_f %.9g dt_c %.9g fmax %.9g dt_pp %.9g dt_pp_ext %.9g | step 2 dt_f %.9g dt_c %.9g fmax %.9g dt_pp %.9g dt_pp_ext %.9g  ghosts %d %d" %
          (rep, o1.dt_f_acc, o1.dt_c_acc, o1.f_force_max, o1.dt_pp_acc, o1.dt_pp_ext_acc, o2.dt_f_acc, o2.dt_c_acc, o2.f_force_max, o2.dt_pp_acc, o2.dt_pp_ext_acc,
           o1.np_ghost, o2.np_ghost), flush=True)
    g.close()
