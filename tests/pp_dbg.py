"""Diagnostic: config-1 clustered PM+PP step on the HIP path against the oracle, particle by particle (which records are off)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import test_gpu_parity as T  # noqa: E402
from cubep3m_amd.particle_mesh import ParticleMesh  # noqa: E402

kw, ic, flags = T.CASES[sys.argv[1] if len(sys.argv) > 1 else "p3m_ext_clustered"]
p = T.cfg1(**kw)
xv = T.make_ic(ic)
xg, pg, xo, po, outs = T.run_step(ParticleMesh, p, xv, (0.005, 0.2, 0.0, 8.0))
d = np.abs(xg[:, 3:] - xo[:, 3:]).max(axis=1)
bad = np.where(~(d <= 1e-4 * np.abs(xo[:, 3:]).max()))[0]
print("n", len(xg), "bad", len(bad), "nan", int(np.isnan(xg[:, 3:]).any(axis=1).sum()))
og, oo = outs[-1]
print("dt_pp_ext", og.dt_pp_ext_acc, oo.dt_pp_ext_acc, "dt_pp", og.dt_pp_acc, oo.dt_pp_acc)
for i in bad[:40]:
    print(i, xg[i, :3], xg[i, 3:], xo[i, 3:])
if len(bad):
    c = np.floor(xg[bad, :3]).astype(int)
    print("cells z:", np.unique(c[:, 2])[:40]); print("cells y:", np.unique(c[:, 1])[:40]); print("cells x:", np.unique(c[:, 0])[:40])
