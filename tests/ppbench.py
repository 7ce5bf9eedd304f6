"""Measurement tool: the two PP kernels on their own (p3m_hip_time_pp), for rocprofv3.
    python3 tests/ppbench.py [uniform|clustered|dense] [reps] [cfg3|big] [steady]
"steady": one whole step first, so that the arrival order (the order velocities are stored in) is the previous step's sorted order, as
in every step of a run but the first; without it the velocities are reached in the upload's random order."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402
from cubep3m_amd.kernels import default_tables  # noqa: E402
from cubep3m_amd.params import Params  # noqa: E402
from cubep3m_amd.particle_mesh import ParticleMesh  # noqa: E402

ic = sys.argv[1] if len(sys.argv) > 1 else "uniform"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
geo = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
if geo == "big":
    p = Params(tiles_node_dim=1, nf_tile=560, ngp=True, ppint=True, pp_ext=True, density_buffer=1.3)
    nside, box = 256, 512.0
else:
    p = Params(**bench.CONFIGS["cfg3"]["params"])
    nside, box = 128, 256.0
fine, coarse = default_tables()
if ic == "uniform":
    xv = bench.make_particles(nside, box)
elif ic == "clustered":
    xv = bench.clustered(nside, box, 2024, 0.3, 3072 * (nside // 128) ** 3, 0.6)
else:
    xv = bench.clustered(nside, box, 2024, 0.3, 48, 0.6)
g = ParticleMesh(p, fine, coarse)
g.upload_particles(xv)
if len(sys.argv) > 4 and sys.argv[4] == "steady":
    g.particle_mesh(0.5, 0.0, 0.0, 8.0)
    g.update_position(0.0, 0.0)
g.link_list_and_pass()
ms_i, ms_e, n_i, n_e = g.time_pp(0.5, 0.0, 8.0, reps=reps)
print("%s %s: intra %.3f ms (%d evaluations, %.3g /s)   extended %.3f ms (%d evaluations, %.3g /s)" %
      (geo, ic, ms_i, n_i, n_i / (ms_i * 1e-3 + 1e-30), ms_e, n_e, n_e / (ms_e * 1e-3 + 1e-30)))
