"""Pins the oracle's FFT-free stages against the REFERENCE'S OWN OBJECT CODE (oracle/_ref, built
where the sources lie by oracle/build_ref.sh): drift, chaining mesh, ghost pass, fine NGP/CIC
deposit, coarse CIC deposit, force halo, coarse max-dt, coarse kick, ghost deletion.  Because the
oracle walks the same linked lists in the same order, agreement is required BIT FOR BIT.
Skipped where oracle/_ref is absent (it is built in the dev container and travels to the GPU box).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import ref_lib
from common import cfg1, clustered_particles, uniform_particles
from ref_stage_run import run_stages, synth_force_c

pytestmark = pytest.mark.ref


def _big_stack():
    # the Fortran objects put whole-array temporaries on the stack (the reference asks for
    # `ulimit -s unlimited`, batch/*.csh); MPI ranks and the 1-rank child both need it
    import resource

    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


CHILD_ENV = dict(os.environ, OMP_NUM_THREADS="1", OMP_STACKSIZE="512M")
HERE = os.path.dirname(os.path.abspath(__file__))
SCAL = (0.05, 0.7, 0.5, 8.0)  # a_mid, dt, dt_old, mass_p
TILES = [(0, 0, 0), (1, 0, 1), (1, 1, 1)]


def make_input(n, box, seed):
    xv = clustered_particles(n, box, seed=seed, frac=0.25, nblobs=12, sigma=1.0, vel_sigma=1.5)
    pid = np.arange(1, n + 1, dtype=np.int64) * 7 + 3
    return xv, pid


@pytest.mark.skipif(not ref_lib.available("cfg1_1rank"), reason="oracle/_ref not built")
def test_single_rank_stages_bitwise():
    # the reference keeps its state in process-global COMMON blocks: run it in a child process
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from ref_lib import Ref
from ref_stage_run import run_stages
d = np.load(sys.argv[1])
ref = Ref('cfg1_1rank')
res = run_stages(ref, d['xv_0'], d['pid_0'], tuple(float(v) for v in d['scal']), [tuple(t) for t in d['tiles']])
np.savez(sys.argv[2], **res)
""" % (HERE, os.path.dirname(HERE))
    xv, pid = make_input(6000, 64.0, 4242)
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), xv_0=xv, pid_0=pid, scal=np.asarray(SCAL, np.float32), tiles=np.asarray(TILES, np.int32))
        subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "in.npz"), os.path.join(td, "out.npz")],
                              stdout=subprocess.DEVNULL, preexec_fn=_big_stack, env=CHILD_ENV)
        ref = dict(np.load(os.path.join(td, "out.npz")))
    compare_with_oracle(cfg1(), [(xv, pid)], [ref])


def compare_with_oracle(p, parts, refs, pids_travel=True):
    a_mid, dt, dt_old, mass_p = SCAL
    nr = len(parts)
    for ngp in (True, False):
        p.ngp = ngp
        o = ol.Oracle(p)
        for r, (xv, pid) in enumerate(parts):
            o.set_particles(r, xv, pid)
        o.update_position(dt, dt_old)
        o.link_list()
        for r in range(nr):
            x, q = o.get_particles(r)
            assert np.array_equal(x, refs[r]["xv_linked"]) and np.array_equal(q, refs[r]["pid_linked"])
        assert o.particle_pass() == 0
        for r in range(nr):
            x, q = o.get_particles(r)
            assert x.shape == refs[r]["xv_passed"].shape, (r, x.shape, refs[r]["xv_passed"].shape)
            assert np.array_equal(x, refs[r]["xv_passed"]), "ghost pass: positions/velocities/order differ"
            if pids_travel:  # the 8-rank reference build has PID_FLAG off (see oracle/build_ref.sh)
                assert np.array_equal(q, refs[r]["pid_passed"])
            for t in TILES:
                key = ("rho_ngp_%d%d%d" if ngp else "rho_cic_%d%d%d") % t
                assert np.array_equal(o.tile_density(r, t, mass_p), refs[r][key]), key
        o.coarse_density(mass_p)
        for r in range(nr):
            assert np.array_equal(o.rho_c(r), refs[r]["rho_c"])
    # halo / max-dt / kick / delete: the same synthetic interior force the reference run was given
    # (ref_stage_run.synth_force_c stands where the FFT result would be), assembled globally.
    nd, ncn, nc = p.nodes_dim, p.nc_node_dim, p.nc_dim
    fg = np.zeros((nc, nc, nc, 3), np.float32)
    for r in range(nr):
        c1, c2, c3 = r // (nd * nd), (r // nd) % nd, r % nd
        fg[c1 * ncn:(c1 + 1) * ncn, c2 * ncn:(c2 + 1) * ncn, c3 * ncn:(c3 + 1) * ncn] = synth_force_c(ncn, r)
    o.distribute_force(fg)
    for r in range(nr):
        assert np.array_equal(o.force_c(r), refs[r]["force_c_halo"]), "coarse force halo"
    o.coarse_max_dt_and_velocity(a_mid, dt)
    assert o.step_out().dt_c_acc == refs[0]["dt_c_acc"]
    for r in range(nr):
        assert np.array_equal(o.get_particles(r)[0], refs[r]["xv_kicked"]), "coarse kick"
    o.delete_particles()
    for r in range(nr):
        x, q = o.get_particles(r)
        assert np.array_equal(x, refs[r]["xv_final"]), "delete_particles order/content"
        if pids_travel:
            assert np.array_equal(q, refs[r]["pid_final"])


@pytest.mark.skipif(not ref_lib.available("cfg1_8rank"), reason="oracle/_ref not built")
def test_eight_rank_ghost_pass_and_deposits_bitwise():
    p = cfg1(nodes_dim=2)
    box = 128.0
    xv, pid = make_input(20000, box, 777)
    Nn = p.nf_physical_node_dim
    parts = []
    # rank = c1*4 + c2*2 + c3 with x <-> c3, y <-> c2, z <-> c1 (mpi_initialization.f90:60-64)
    for rk in range(8):
        c1, c2, c3 = rk // 4, (rk // 2) % 2, rk % 2
        lo = np.array([c3, c2, c1], np.float32) * Nn
        m = np.all((xv[:, :3] >= lo) & (xv[:, :3] < lo + Nn), axis=1)
        loc = xv[m].copy()
        loc[:, :3] -= lo
        parts.append((loc, pid[m]))
    with tempfile.TemporaryDirectory() as td:
        d = {"scal": np.asarray(SCAL, np.float32), "tiles": np.asarray(TILES, np.int32)}
        for rk, (a, b) in enumerate(parts):
            d["xv_%d" % rk] = a
            d["pid_%d" % rk] = b
        np.savez(os.path.join(td, "in.npz"), **d)
        subprocess.check_call(["/opt/conda/bin/mpiexec", "-n", "8", sys.executable, os.path.join(HERE, "ref_stage_run.py"),
                               "cfg1_8rank", os.path.join(td, "in.npz"), td], stdout=subprocess.DEVNULL, env=CHILD_ENV, preexec_fn=_big_stack, timeout=600)
        refs = [dict(np.load(os.path.join(td, "ref_out_%d.npz" % rk))) for rk in range(8)]
    for rk in range(8):
        assert list(refs[rk]["cart_coords"]) == [rk // 4, (rk // 2) % 2, rk % 2]
    compare_with_oracle(p, parts, refs, pids_travel=False)
