"""TEST INFRASTRUCTURE (dev container only): lets the REFERENCE'S OWN projection.o (oracle/_ref/<cfg>/libref.so) project a
seeded particle set and write its three projection files.
usage: python ref_proj_run.py <cfg> <out.npz>      (child process of tests/golden/make_ref_projection.py)"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_lib  # noqa: E402
from common import clustered_particles  # noqa: E402
from cubep3m_amd import io_formats as iof  # noqa: E402

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
RV = np.asarray([0.4375, 8.0, 1.286], np.float32)   # a, mass_p, z_projection


def main():
    cfg, out = sys.argv[1:3]
    ref = ref_lib.Ref(cfg)
    box = float(ref.nf_physical_node_dim)
    n = 1500                                           # the reference builds are small (max_np, max_buf): see oracle/build_ref.sh
    xv = clustered_particles(n, box, seed=33, frac=0.25, nblobs=6, sigma=1.0, vel_sigma=0.5)
    pid = np.arange(1, n + 1, dtype=np.int64)
    ref.set_scalars(float(RV[0]), 0.0, 0.0, float(RV[1]))
    ref.set_particles(xv, pid)
    ref.link_list()
    ref.particle_pass()
    Np = ref.nf_physical_node_dim * ref.nodes_dim
    maps = [np.zeros((Np, Np), np.float32) for _ in range(3)]
    L = ref.L
    L.ref_projection.argtypes = [f32p] * 4
    L.ref_projection(RV, *maps)
    d = os.path.dirname(ref_lib.so_path(cfg))
    res = dict(xv=xv, pid=pid, rv=RV, pxy=maps[0], pxz=maps[1], pyz=maps[2])
    for ax, name in zip(("xy", "xz", "yz"), iof.projection_names(float(RV[2]))):
        res["file_" + ax] = np.fromfile(os.path.join(d, name), np.uint8)
        res["name_" + ax] = np.asarray(name)
    np.savez(out, **res)
    L.ref_finalize()


if __name__ == "__main__":
    main()
