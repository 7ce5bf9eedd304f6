"""Whole-step and stage parity at BASELINE's FULL sizes: the HIP path (through the C ABI) against
the CPU oracle on the same seeded inputs at configs 2 and 3 (256^3 fine mesh / 128^3 particles, 2^3 tiles of 176 cells:
particle_mesh_threaded.f90:85-368), the CIC deposit and gather at 176- and 560-cell rows (fine_cic_mass.f90:13-43), and the
slab / pencil coarse mesh at the bench's own size (nc = 256, nc_slab = 32: fftw3ds.f90:103-183).
Bars: particle set exact, positions bit-identical (no drift), kick rel. rms <= 1e-5 matched by PID,
dt limits rel. 1e-5."""
import numpy as np
import pytest

import oracle_lib as ol
from common import COARSE_TABLE, FINE_TABLE, by_pid, clustered_particles, grid_jitter_particles, rel_rms
from cubep3m_amd.params import Params

pytestmark = pytest.mark.gpu

KICK_TOL = 1e-5
DT_TOL = 1e-5


@pytest.fixture(scope="module")
def PM():
    from cubep3m_amd.particle_mesh import ParticleMesh

    return ParticleMesh


def appendix_c_clustered(n, box, seed=2024):
    """SURVEY Appendix C's clustered input at its own density: 30 % of the particles in Gaussian blobs of sigma = 0.6 cells,
    48 blobs per 32 768 particles (about 205 members each), the rest uniform."""
    return clustered_particles(n, box, seed=seed, frac=0.3, nblobs=max(1, (48 * n) // 32768), sigma=0.6)


@pytest.mark.parametrize("ngp,pp", [(True, False), (True, True), (False, False), (False, True)],
                         ids=["cfg2_ngp", "cfg3_ngp_pp_ext", "cfg2_cic", "cfg3_cic_pp_ext"])
def test_configs_2_and_3_whole_step_at_full_size(PM, ngp, pp):
    """BASELINE config 2 (PM only) and config 3 (PM + PPINT + PP_EXT) at 256^3 cells / 128^3 particles, 2^3 tiles of 176,
    with the NGP and with the CIC fine mesh: one whole `particle_mesh` step on the GPU and on the oracle."""
    # -DPPINT only exists inside the NGP branch (particle_mesh_threaded.f90:260-287); the CIC build keeps -DPP_EXT (:378-624)
    p = Params(tiles_node_dim=2, nf_tile=176, ngp=ngp, ppint=pp and ngp, pp_ext=pp, density_buffer=1.5, cores=8)
    n = 128 ** 3
    xv = appendix_c_clustered(n, 256.0) if pp else grid_jitter_particles(128, 256.0, seed=778, sigma=0.4)
    g = PM(p, FINE_TABLE, COARSE_TABLE)
    g.upload_particles(xv)
    og = g.particle_mesh(0.005, 0.2, 0.0, 8.0)
    xg, pg = by_pid(*g.download_particles())
    del g
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    o.set_particles(0, xv)
    oo = o.particle_mesh(0.005, 0.2, 0.0, 8.0)
    xo, po = by_pid(*o.get_particles(0))
    o.close()
    assert og.np_total == oo.np_total == n and og.np_ghost == oo.np_ghost and og.np_deleted == oo.np_deleted == 0
    assert np.array_equal(pg, po)
    assert np.array_equal(xg[:, :3], xo[:, :3])                   # v = 0, dt_old = 0: positions untouched on both sides
    err = rel_rms(xg[:, 3:], xo[:, 3:])
    assert err <= KICK_TOL, err
    names = ("dt_f_acc", "dt_c_acc") + (("dt_pp_ext_acc",) if pp else ()) + (("dt_pp_acc",) if pp and ngp else ())
    for name in names:
        assert getattr(og, name) == pytest.approx(getattr(oo, name), rel=DT_TOL), name
    assert og.sum_rho_f == pytest.approx(oo.sum_rho_f, rel=1e-6) and og.sum_rho_c == pytest.approx(oo.sum_rho_c, rel=1e-6)


def test_cic_deposit_at_the_bench_tile_size(PM):
    """fine_cic_mass.f90:13-43 on 560-cell rows (the headline's tile): the CIC density of a one-million-particle clustered
    sample, ghosts included, cell by cell against the oracle."""
    p = Params(tiles_node_dim=1, nf_tile=560, ngp=False, density_buffer=0.1)
    xv = clustered_particles(1_000_000, 512.0, seed=56, frac=0.4, nblobs=2000, sigma=1.2)
    g = PM(p, FINE_TABLE, COARSE_TABLE)
    o = ol.Oracle(p)
    g.upload_particles(xv)
    o.set_particles(0, xv)
    g.link_list_and_pass()
    o.link_list()
    assert o.particle_pass() == 0
    rg, ro = g.tile_density((0, 0, 0), 8.0), o.tile_density(0, (0, 0, 0), 8.0)
    assert rg.shape == ro.shape
    assert np.abs(rg - ro).max() <= 4e-6 * max(1.0, float(np.abs(ro).max()))
    assert float(rg.sum(dtype=np.float64)) == pytest.approx(float(ro.sum(dtype=np.float64)), rel=1e-7)
    assert np.all(rg[:, :, 560:] == 0)


@pytest.mark.parametrize("pencil", [False, True])
def test_distributed_coarse_mesh_at_the_bench_size(pencil):
    """The coarse path the default bench runs -- 2x2x2 ranks of ONE 560-cell tile each, nc = 256, nc_slab = 32 (slabs), and
    the pencil twin: cube<->slab redistribution, the distributed transform with its all-to-all, the kernel multiply, the
    force halo -- rank by rank against the oracle's coarse_density / coarse_force."""
    from cubep3m_amd.group import ParticleMeshGroup

    p = Params(nodes_dim=2, tiles_node_dim=1, nf_tile=560, ngp=True, lrckcorr=True, pencil=pencil, density_buffer=0.05, cores=1)
    assert p.nc_dim == 256
    box = float(p.nf_physical_dim)
    xv = clustered_particles(400_000, box, seed=5, frac=0.3, nblobs=400, sigma=3.0, vel_sigma=0.5)
    pid = np.arange(1, len(xv) + 1, dtype=np.int64)
    g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE)
    parts = g.scatter_global(xv, pid)
    o = ol.Oracle(p)
    o.L.orc_coarse_kernel(o.h, np.ascontiguousarray(COARSE_TABLE, np.float32))     # the 560^3 fine kernel is not needed here
    for r in range(8):
        o.set_particles(r, *parts[r])
    out = g.particle_mesh(0.01, 0.0, 0.0, 8.0)     # dt = dt_old = 0: leaves the cell-sorted records (with ghosts) for the probe
    assert out.np_total == len(xv)
    o.link_list()
    assert o.particle_pass() == 0
    o.coarse_density(8.0)
    o.coarse_force()
    for i in range(8):
        rg, fg = g.coarse(8.0, i)
        ro, fo = o.rho_c(i), o.force_c(i)
        assert np.abs(rg - ro).max() <= 4e-6 * np.abs(ro).max(), i
        assert rel_rms(fg, fo) < 3e-6, i
    g.close()
