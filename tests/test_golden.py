"""Oracle vs the committed golden fixtures (tests/golden/ref_stages_*.npz), which hold outputs of the
REFERENCE'S OWN OBJECT CODE for the FFT-free stages (generator: tests/golden/make_ref_fixtures.py).
Runs anywhere (no /root/reference, no oracle/_ref needed).  Bit-exact."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from common import cfg1
from ref_stage_run import synth_force_c

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dense(d, key, shape):
    a = np.zeros(int(np.prod(shape)), np.float32)
    a[d[key + "_idx"]] = d[key + "_val"]
    return a.reshape(shape)


@pytest.mark.parametrize("nd", [1, 2])
def test_oracle_reproduces_reference_stage_outputs(nd):
    d = np.load(os.path.join(G, "ref_stages_%drank.npz" % (nd ** 3)))
    a_mid, dt, dt_old, mass_p = (float(v) for v in d["scal"])
    tiles = [tuple(int(x) for x in t) for t in d["tiles"]]
    nr = nd ** 3
    key = (lambda r, k: k) if nd == 1 else (lambda r, k: "r%d_%s" % (r, k))
    for ngp in (True, False):
        p = cfg1(nodes_dim=nd, ngp=ngp)
        o = ol.Oracle(p)
        for r in range(nr):
            o.set_particles(r, d[key(r, "xv_in")], d[key(r, "pid_in")])
        o.update_position(dt, dt_old)
        o.link_list()
        assert o.particle_pass() == 0
        nf = p.nf_tile
        for r in range(nr):
            x, q = o.get_particles(r)
            assert np.array_equal(x, d[key(r, "xv_passed")])
            if nd == 1:
                assert np.array_equal(q, d["pid_passed"])
            for t in tiles:
                name = ("rho_ngp_%d%d%d" if ngp else "rho_cic_%d%d%d") % t
                assert np.array_equal(o.tile_density(r, t, mass_p), dense(d, key(r, name), (nf, nf, nf + 2)))
        o.coarse_density(mass_p)
        for r in range(nr):
            assert np.array_equal(o.rho_c(r), d[key(r, "rho_c")])
    ncn, nc = p.nc_node_dim, p.nc_dim
    fg = np.zeros((nc, nc, nc, 3), np.float32)
    for r in range(nr):
        c1, c2, c3 = r // (nd * nd), (r // nd) % nd, r % nd
        fg[c1 * ncn:(c1 + 1) * ncn, c2 * ncn:(c2 + 1) * ncn, c3 * ncn:(c3 + 1) * ncn] = synth_force_c(ncn, r)
    o.distribute_force(fg)
    for r in range(nr):
        assert np.array_equal(o.force_c(r), d[key(r, "force_c_halo")])
    o.coarse_max_dt_and_velocity(a_mid, dt)
    assert o.step_out().dt_c_acc == d[key(0, "dt_c_acc")]
    for r in range(nr):
        assert np.array_equal(o.get_particles(r)[0], d[key(r, "xv_kicked")])
    o.delete_particles()
    for r in range(nr):
        x, q = o.get_particles(r)
        assert np.array_equal(x, d[key(r, "xv_final")])
        if nd == 1:
            assert np.array_equal(q, d["pid_final"])


@pytest.mark.parametrize("cfg,ngp,ppint", [("cfg1_pp", True, True), ("cfg1_1rank", True, False), ("cfg1_cic", False, False)])
def test_oracle_reproduces_reference_fine_velocity(cfg, ngp, ppint):
    """fine_velocity.f90 (force maximum, NGP / CIC gather + kick, intra-cell PP: SURVEY 8a rows a10-a12) as the
    reference's object code computed it on a synthetic force box (generator tests/golden/make_ref_fine_velocity.py)."""
    from test_oracle_vs_ref import oracle_fine_velocity

    d = np.load(os.path.join(G, "ref_fine_velocity.npz"))
    xv, pid = d["xv_in"], d["pid_in"]
    fmax2, ppmax, x, q = oracle_fine_velocity(cfg1(ngp=ngp, ppint=ppint), xv, pid, tuple(float(v) for v in d["scal"]))
    n = len(xv)
    assert len(x) == int(d[cfg + "_np_passed"])
    assert np.array_equal(q[:n], pid) and np.array_equal(x[:n, :3], xv[:, :3])
    assert np.array_equal(x[:n, 3:], d[cfg + "_vel"])
    assert np.array_equal(np.sqrt(fmax2), d[cfg + "_f_force_max"])
    assert np.array_equal(ppmax, d[cfg + "_pp_force_max"])
    assert (ppmax.max() > 10.0) == ppint


def test_oracle_reproduces_reference_coarse_ngp_build():
    """-DCOARSE_NGP (coarse_cic_mass.f90:21-24, coarse_cic_mass_buffer.f90:26-29, coarse_velocity.f90:146-149): coarse
    density and coarse kick of the reference's object code built with the switch (tests/golden/make_ref_coarse_ngp.py)."""
    d = np.load(os.path.join(G, "ref_coarse_ngp.npz"))
    a_mid, dt, dt_old, mass_p = (float(v) for v in d["scal"])
    p = cfg1(coarse_ngp=True)
    o = ol.Oracle(p)
    o.set_particles(0, d["xv_in"], d["pid_in"])
    o.update_position(dt, dt_old)
    o.link_list()
    assert o.particle_pass() == 0
    assert np.array_equal(o.get_particles(0)[0], d["xv_passed"])
    o.coarse_density(mass_p)
    assert np.array_equal(o.rho_c(0), d["rho_c"])
    plain = ol.Oracle(cfg1())                                    # the switch does change the answer
    plain.set_particles(0, d["xv_in"], d["pid_in"])
    plain.update_position(dt, dt_old)
    plain.link_list()
    plain.particle_pass()
    plain.coarse_density(mass_p)
    assert not np.array_equal(plain.rho_c(0), d["rho_c"])
    ncn = p.nc_node_dim
    o.distribute_force(synth_force_c(ncn, 0).reshape(ncn, ncn, ncn, 3))
    assert np.array_equal(o.force_c(0), d["force_c_halo"])
    o.coarse_max_dt_and_velocity(a_mid, dt)
    assert o.step_out().dt_c_acc == d["dt_c_acc"]
    assert np.array_equal(o.get_particles(0)[0], d["xv_kicked"])
    o.delete_particles()
    x, q = o.get_particles(0)
    assert np.array_equal(x, d["xv_final"]) and np.array_equal(q, d["pid_final"])
