import sys, json
sys.path.insert(0,'.')
from cubep3m_amd.params import Params
from cubep3m_amd.particle_mesh import ParticleMesh
import bench
for name in sys.argv[1:]:
    cfg = bench.CONFIGS[name]; p = Params(**cfg["params"])
    pm = ParticleMesh(p)
    xv = bench.make_particles(cfg["nside_rank"], float(p.nf_physical_node_dim)); pm.upload_particles(xv)
    pm.particle_mesh(0.5,0.05,0.05,8.0)
    res = {n: round(pm.time_fft_pass(i, 20)[0],4) for i,n in enumerate(pm.FFT_PASSES)}
    print(name, res, 'sweep', round(pm.time_fine_sweep(8.0,5),3), flush=True)
