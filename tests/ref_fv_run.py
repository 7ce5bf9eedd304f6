"""TEST INFRASTRUCTURE: runs the reference's fine_velocity.f90 (oracle/_ref, the non-inlined twin of
particle_mesh_threaded.f90:208-368: force maximum, NGP/CIC gather + kick, intra-cell PP) on a seeded input and a
synthetic force box standing where the FFT result would be.  One process = one reference rank (COMMON blocks).

    python tests/ref_fv_run.py cfg1_pp in.npz out.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def synth_force_f(fb, tile, seed=1234):
    """A deterministic force box [k][j][i][3] of fb^3 cells standing in for the three inverse transforms."""
    rng = np.random.default_rng(seed + 100 * tile[2] + 10 * tile[1] + tile[0])
    return (rng.standard_normal((fb, fb, fb, 3)) * 3.0).astype(np.float32)


def all_tiles(T):
    return [(i, j, k) for k in range(T) for j in range(T) for i in range(T)]


def run_fine_velocity(ref, xv, pid, scal):
    a_mid, dt, dt_old, mass_p = scal
    out = {}
    ref.set_scalars(a_mid, dt, dt_old, mass_p)
    ref.set_particles(xv, pid)
    ref.link_list()
    ref.particle_pass()
    fb = ref.nf_tile - 2 * ref.nf_buf + 3
    fmax, ppmax = [], []
    for t in all_tiles(ref.tiles_node_dim):
        a, b = ref.fine_velocity(t, synth_force_f(fb, t))
        fmax.append(a)
        ppmax.append(b)
    out["f_force_max"] = np.asarray(fmax, np.float32)
    out["pp_force_max"] = np.asarray(ppmax, np.float32)
    out["xv_kicked"], out["pid_kicked"] = ref.get_particles()
    return out


if __name__ == "__main__":
    from ref_lib import Ref

    cfg, inp, outp = sys.argv[1:4]
    ref = Ref(cfg)
    d = np.load(inp)
    res = run_fine_velocity(ref, d["xv"], d["pid"], tuple(float(v) for v in d["scal"]))
    np.savez(outp, **res)
