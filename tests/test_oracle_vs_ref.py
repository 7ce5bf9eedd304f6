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


@pytest.mark.parametrize("cfg,kw", [("cfg1_1rank", {}), ("cfg1_cngp", dict(coarse_ngp=True))])
def test_single_rank_stages_bitwise(cfg, kw):
    if not ref_lib.available(cfg):
        pytest.skip("oracle/_ref/%s not built" % cfg)
    # the reference keeps its state in process-global COMMON blocks: run it in a child process
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from ref_lib import Ref
from ref_stage_run import run_stages
d = np.load(sys.argv[1])
ref = Ref(%r)
res = run_stages(ref, d['xv_0'], d['pid_0'], tuple(float(v) for v in d['scal']), [tuple(t) for t in d['tiles']])
np.savez(sys.argv[2], **res)
""" % (HERE, os.path.dirname(HERE), cfg)
    xv, pid = make_input(6000, 64.0, 4242)
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), xv_0=xv, pid_0=pid, scal=np.asarray(SCAL, np.float32), tiles=np.asarray(TILES, np.int32))
        subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "in.npz"), os.path.join(td, "out.npz")],
                              stdout=subprocess.DEVNULL, preexec_fn=_big_stack, env=CHILD_ENV)
        ref = dict(np.load(os.path.join(td, "out.npz")))
    compare_with_oracle(cfg1(**kw), [(xv, pid)], [ref])


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


# ---- fine_velocity.f90: force maximum, NGP / CIC gather + kick, intra-cell PP -------------------
def fine_velocity_input(n=5000, seed=99):
    # 35 % of the particles in tight blobs: dozens of fine cells hold 2...40 particles (the PPINT pair loops), a few pairs
    # closer than rsoft (the hard cut of :340)
    xv = clustered_particles(n, 64.0, seed=seed, frac=0.35, nblobs=10, sigma=0.5, vel_sigma=0.5)
    xv[1::97, :3] = xv[0:-1:97, :3] + np.float32(0.01)       # pairs inside rsoft = 0.1
    xv[:, :3] = np.clip(xv[:, :3], 0, np.float32(63.999))
    pid = np.arange(1, n + 1, dtype=np.int64) * 3 + 1
    return xv, pid


def oracle_fine_velocity(p, xv, pid, scal):
    from ref_fv_run import all_tiles, synth_force_f

    a_mid, dt, dt_old, mass_p = scal
    o = ol.Oracle(p)
    o.set_particles(0, xv, pid)
    o.link_list()
    assert o.particle_pass() == 0
    fb = p.nf_physical_tile_dim + 3
    fmax2, ppmax = [], []
    for t in all_tiles(p.tiles_node_dim):
        a, b = o.tile_velocity(0, t, synth_force_f(fb, t), a_mid, dt, mass_p)
        fmax2.append(a)
        ppmax.append(b)
    x, q = o.get_particles(0)
    return np.asarray(fmax2, np.float32), np.asarray(ppmax, np.float32), x, q


def check_fine_velocity(p, xv, pid, scal, ref):
    fmax2, ppmax, x, q = oracle_fine_velocity(p, xv, pid, scal)
    assert np.array_equal(q, ref["pid_kicked"])
    assert np.array_equal(x[:, :3], ref["xv_kicked"][:, :3])
    assert np.array_equal(x[:, 3:], ref["xv_kicked"][:, 3:]), "fine kick / intra-cell PP kick differ from fine_velocity.f90"
    # fine_velocity.f90:39-50 keeps max |F|, the threaded file max |F|^2 (:208-223) and takes the root later (:643-652):
    # the correctly rounded square root is monotonic, so the two agree exactly
    assert np.array_equal(np.sqrt(fmax2), ref["f_force_max"])
    assert np.array_equal(ppmax, ref["pp_force_max"])
    return ppmax


@pytest.mark.parametrize("cfg,ngp,ppint", [("cfg1_pp", True, True), ("cfg1_1rank", True, False), ("cfg1_cic", False, False)])
def test_fine_velocity_bitwise(cfg, ngp, ppint):
    if not ref_lib.available(cfg):
        pytest.skip("oracle/_ref not built")
    xv, pid = fine_velocity_input()
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), xv=xv, pid=pid, scal=np.asarray(SCAL, np.float32))
        subprocess.check_call([sys.executable, os.path.join(HERE, "ref_fv_run.py"), cfg, os.path.join(td, "in.npz"), os.path.join(td, "out.npz")],
                              stdout=subprocess.DEVNULL, preexec_fn=_big_stack, env=CHILD_ENV)
        ref = dict(np.load(os.path.join(td, "out.npz")))
    ppmax = check_fine_velocity(cfg1(ngp=ngp, ppint=ppint), xv, pid, SCAL, ref)
    if ppint:
        assert ppmax.max() > 100.0   # the pair loops did run on dense cells
    else:
        assert ppmax.max() == 0.0
