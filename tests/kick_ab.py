"""Diagnostic: two whole steps of a bench config, velocities by PID and the step outputs dumped to an .npz:
    P3M_KICK_UNFUSED=1 python tests/kick_ab.py cfg1 a.npz ; python tests/kick_ab.py cfg1 b.npz ; python tests/kick_ab.py --cmp a.npz b.npz
(the force box + k_fine_kick_rows pair against the fused inverse-x + kick pass, kick_fused.hip)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        x, y = a[k], b[k]
        if x.shape != y.shape:
            print("%-12s shapes differ %s %s" % (k, x.shape, y.shape)); ok = False; continue
        d = np.abs(x.astype(np.float64) - y.astype(np.float64))
        nz = int((x != y).sum())
        rel = float(np.sqrt((d ** 2).mean()) / max(np.sqrt((x.astype(np.float64) ** 2).mean()), 1e-300))
        print("%-12s n=%d differing=%d max|d|=%.3e rel-rms=%.3e" % (k, x.size, nz, d.max() if d.size else 0.0, rel))
        ok = ok and rel < 1e-6
    sys.exit(0 if ok else 1)

from cubep3m_amd.params import Params
from cubep3m_amd.group import ParticleMeshGroup
import bench
name, out = sys.argv[1], sys.argv[2]
ic = sys.argv[3] if len(sys.argv) > 3 else "uniform"
vs = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
cfg = bench.CONFIGS[name]; p = Params(**cfg["params"])
grp = ParticleMeshGroup(p, 0, 1)
nside, box = cfg["nside_rank"], float(p.nf_physical_node_dim)
for i, r in enumerate(grp.local_ranks):
    xv = bench.make_particles(nside, box, seed=12345 + r) if ic == "uniform" else bench.clustered(nside, box, 2024 + r, 0.3, 48 * (nside // 32) ** 3, 0.6)
    xv[:, 3:] = np.random.default_rng(900 + r).normal(0, vs, (len(xv), 3)).astype(np.float32)
    grp.upload_particles(i, xv, np.arange(1, len(xv) + 1, dtype=np.int64) + r * len(xv))
outs = []
for s in range(2):
    o = grp.particle_mesh(0.5, 0.05, 0.05 if s else 0.0, 8.0)
    outs.append([o.dt_f_acc, o.dt_pp_acc, o.dt_pp_ext_acc, o.dt_c_acc, o.f_force_max, o.np_total, o.np_ghost, o.np_deleted])
xs, ps = [], []
for i in range(len(grp.local_ranks)):
    xv, pid = grp.download_particles(i)
    xs.append(xv); ps.append(pid)
xv = np.concatenate(xs); pid = np.concatenate(ps)
o = np.argsort(pid, kind="stable")
np.savez(out, xv=xv[o], pid=pid[o], outs=np.array(outs, np.float64))
print(name, "np", len(pid), "outs", outs, flush=True)
grp.close()
