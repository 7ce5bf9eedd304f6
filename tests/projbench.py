"""Times the density projection (projection.f90) on one rank's share of the default workload: python tests/projbench.py"""
import sys, time
sys.path.insert(0, '.')
from cubep3m_amd.params import Params
from cubep3m_amd.particle_mesh import ParticleMesh
import bench
cfg = bench.CONFIGS["big512"]; p = Params(**cfg["params"])
pm = ParticleMesh(p)
pm.upload_particles(bench.make_particles(cfg["nside_rank"], float(p.nf_physical_node_dim)))
pm.particle_mesh(0.5, 0.05, 0.05, 8.0)
for rep in range(3):
    t0 = time.perf_counter(); pm.link_list_and_pass(); t1 = time.perf_counter()
    pxy, pxz, pyz, tot = pm.projection(8.0); t2 = time.perf_counter()
    pm.delete_particles(); t3 = time.perf_counter()
    print("pass+sort %.2f ms, projection (deposit, 3 maps, download) %.2f ms, delete %.2f ms, mass %.6g" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), tot))
