"""Generates tests/golden/ref_stages_{1rank,8rank}.npz from the REFERENCE'S OWN OBJECT CODE
(oracle/_ref, built by oracle/build_ref.sh from /root/reference/source_threads where it lies).
Run in the dev container only:   python tests/golden/make_ref_fixtures.py
The fixtures hold inputs and the reference's outputs (data only) for the FFT-free stages:
drift -> link_list -> particle_pass -> fine NGP/CIC deposit -> coarse CIC deposit ->
coarse_force_buffer -> coarse_max_dt -> coarse_velocity -> delete_particles.
Dense meshes are stored sparsely (flat index + value of the non-zero cells).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))

from common import cfg1, clustered_particles  # noqa: E402

SCAL = np.asarray((0.05, 0.7, 0.5, 8.0), np.float32)  # a_mid, dt, dt_old, mass_p
TILES = np.asarray([(0, 0, 0), (1, 0, 1)], np.int32)


def big_stack():
    import resource

    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


def sparse(a):
    f = a.ravel()
    idx = np.flatnonzero(f)
    return idx.astype(np.int32), f[idx]


def pack(res, keep_dense=("rho_c", "force_c_halo")):
    out = {}
    for k, v in res.items():
        if k.startswith("rho_ngp") or k.startswith("rho_cic"):
            out[k + "_idx"], out[k + "_val"] = sparse(v)
        elif k in ("hoc", "ll", "xv_linked", "pid_linked"):
            continue
        else:
            out[k] = v
    return out


def main():
    env = dict(os.environ, OMP_NUM_THREADS="1", OMP_STACKSIZE="512M")
    worker = os.path.join(TESTS, "ref_stage_run.py")
    # ---- one rank ---------------------------------------------------------------------------
    xv = clustered_particles(1200, 64.0, seed=31, frac=0.25, nblobs=6, sigma=1.0, vel_sigma=1.5)
    pid = np.arange(1, 1201, dtype=np.int64) * 7 + 3
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), xv_0=xv, pid_0=pid, scal=SCAL, tiles=TILES)
        subprocess.check_call(["/opt/conda/bin/mpiexec", "-n", "1", sys.executable, worker, "cfg1_1rank", os.path.join(td, "in.npz"), td],
                              env=env, preexec_fn=big_stack, stdout=subprocess.DEVNULL)
        res = dict(np.load(os.path.join(td, "ref_out_0.npz")))
    out = pack(res)
    out.update(xv_in=xv, pid_in=pid, scal=SCAL, tiles=TILES)
    np.savez_compressed(os.path.join(HERE, "ref_stages_1rank.npz"), **out)
    # ---- 2x2x2 ranks -------------------------------------------------------------------------
    p = cfg1(nodes_dim=2)
    Nn = p.nf_physical_node_dim
    xv = clustered_particles(1600, 128.0, seed=32, frac=0.25, nblobs=8, sigma=1.0, vel_sigma=1.5)
    pid = np.arange(1, 1601, dtype=np.int64) * 5 + 1
    d = {"scal": SCAL, "tiles": TILES}
    for rk in range(8):
        c1, c2, c3 = rk // 4, (rk // 2) % 2, rk % 2
        lo = np.array([c3, c2, c1], np.float32) * Nn
        m = np.all((xv[:, :3] >= lo) & (xv[:, :3] < lo + Nn), axis=1)
        loc = xv[m].copy()
        loc[:, :3] -= lo
        d["xv_%d" % rk], d["pid_%d" % rk] = loc, pid[m]
    out = {"scal": SCAL, "tiles": TILES}
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), **d)
        subprocess.check_call(["/opt/conda/bin/mpiexec", "-n", "8", sys.executable, worker, "cfg1_8rank", os.path.join(td, "in.npz"), td],
                              env=env, preexec_fn=big_stack, stdout=subprocess.DEVNULL)
        for rk in range(8):
            res = pack(dict(np.load(os.path.join(td, "ref_out_%d.npz" % rk))))
            for k, v in res.items():
                if k in ("pid_passed", "pid_final", "pid_linked"):  # PID_FLAG off in the 8-rank build
                    continue
                out["r%d_%s" % (rk, k)] = v
            out["r%d_xv_in" % rk], out["r%d_pid_in" % rk] = d["xv_%d" % rk], d["pid_%d" % rk]
    np.savez_compressed(os.path.join(HERE, "ref_stages_8rank.npz"), **out)
    for f in ("ref_stages_1rank.npz", "ref_stages_8rank.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
