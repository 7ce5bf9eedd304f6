"""TEST INFRASTRUCTURE: runs the reference's FFT-free stages (oracle/_ref) on a seeded input and
returns / dumps every intermediate.  Used in-process for 1 rank and under `mpiexec -n 8` for the
2x2x2 decomposition (each process is one reference rank).

    mpiexec -n 8 python tests/ref_stage_run.py cfg1_8rank in.npz outdir
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def synth_force_c(ncn, rank, seed=99):
    """A deterministic smooth+noise interior coarse force standing in for the FFT result."""
    rng = np.random.default_rng(seed + rank)
    return rng.standard_normal((ncn, ncn, ncn, 3)).astype(np.float32)


def run_stages(ref, xv, pid, scal, tiles):
    a_mid, dt, dt_old, mass_p = scal
    out = {}
    ref.set_scalars(a_mid, dt, dt_old, mass_p)
    ref.set_particles(xv, pid)
    ref.update_position()
    ref.link_list()
    out["xv_linked"], out["pid_linked"] = ref.get_particles()
    ref.particle_pass()
    out["xv_passed"], out["pid_passed"] = ref.get_particles()
    hoc, ll = ref.get_lists()
    out["hoc"], out["ll"] = hoc, ll
    for t in tiles:
        out["rho_ngp_%d%d%d" % tuple(t)] = ref.fine_deposit(t, True)
        out["rho_cic_%d%d%d" % tuple(t)] = ref.fine_deposit(t, False)
    out["rho_c"] = ref.coarse_mass()
    f = synth_force_c(ref.nc_node_dim, ref.rank)
    ref.set_force_c(f)
    ref.coarse_force_buffer()
    out["force_c_halo"] = ref.get_force_c()
    out["dt_c_acc"] = np.float32(ref.coarse_max_dt())
    ref.coarse_velocity()
    out["xv_kicked"], _ = ref.get_particles()
    ref.delete_particles()
    out["xv_final"], out["pid_final"] = ref.get_particles()
    return out


if __name__ == "__main__":
    from ref_lib import Ref

    cfg, inp, outdir = sys.argv[1:4]
    ref = Ref(cfg)
    d = np.load(inp)
    r = ref.rank
    res = run_stages(ref, d["xv_%d" % r], d["pid_%d" % r], tuple(float(v) for v in d["scal"]), [tuple(t) for t in d["tiles"]])
    res["cart_coords"] = np.asarray(ref.cart_coords, np.int32)
    res["neighbors"] = ref.neighbors()
    np.savez(os.path.join(outdir, "ref_out_%d.npz" % r), **res)
    ref.L.ref_finalize()
