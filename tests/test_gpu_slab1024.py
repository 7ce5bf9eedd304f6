"""BASELINE config 4's literal 1024^3 coarse mesh (SURVEY section 8 config note (ii); fftw3ds.f90:103-183 at nc_dim = 1024,
nc_slab = 128, cubepm.par:196-197): the distributed slab transform of eight logical ranks on a P3M_FLAG_COARSE_ONLY group --
cube -> slab redistribution, x and y passes, the all-to-all transpose, z pass -- against the oracle's FFT, and coarse_force
(coarse_force.f90:18-90: multiply, three inverse transforms, slab -> cube, halo) against closed forms of the real-space
kernel (kernel_initialization.f90:293-336, :366-457) plus translation invariance."""
import numpy as np
import pytest

import oracle_lib as ol
from common import COARSE_TABLE
from cubep3m_amd.group import rank_coords
from cubep3m_amd.params import Params

P3M_ESTATE = -5   # include/p3m_hip.h:70

pytestmark = pytest.mark.gpu


def make_group(nc_rank_tiles=4, pencil=False):
    from cubep3m_amd.group import ParticleMeshGroup

    p = Params(nodes_dim=2, tiles_node_dim=nc_rank_tiles, nf_tile=560, coarse_only=True, pencil=pencil, cores=1)
    g = ParticleMeshGroup(p, 0, 1, set_kernels=False)
    assert g.nlocal == 8
    return p, g


def scatter(g, p, field):
    n = p.nc_node_dim
    for i, r in enumerate(g.local_ranks):
        c1, c2, c3 = rank_coords(r, p.nodes_dim)
        g.set_coarse_density(i, field[c1 * n:(c1 + 1) * n, c2 * n:(c2 + 1) * n, c3 * n:(c3 + 1) * n])


@pytest.mark.parametrize("tiles", [1, 4], ids=["nc256", "nc1024"])
def test_distributed_forward_transform_vs_oracle_fft(tiles):
    p, g = make_group(tiles)
    nc = p.nc_dim
    assert nc == 256 * tiles and p.nc_slab == nc // 8
    rng = np.random.default_rng(nc)
    field = np.zeros((nc, nc, nc + 2), np.float32)
    k = 40 * nc * nc // 256
    idx = rng.integers(0, nc, (k, 3))
    np.add.at(field, (idx[:, 0], idx[:, 1], idx[:, 2]), (rng.random(k, dtype=np.float32) * 8.0 + 1.0).astype(np.float32))
    scatter(g, p, field[:, :, :nc])
    g.coarse_transform("forward")
    ref = ol.fft3d(field, nc, +1)                               # in place: [kz][ky][kx] interleaved re/im
    refc = ref.view(np.complex64)                               # [kz][ky][nc/2+1]
    scale = float(np.abs(refc).max())
    hx = nc // 2 + 1
    for i, r in enumerate(g.local_ranks):
        hat = g.coarse_hat(i)                                   # [local ky][kz][kx padded]
        ky0 = r * p.nc_slab
        want = refc[:, ky0:ky0 + p.nc_slab, :].transpose(1, 0, 2)
        assert np.abs(hat[:, :, :hx] - want).max() / scale < 2e-6, r
        assert np.all(hat[:, :, hx:] == 0), r
    assert g.coarse_exchange_bytes == p.nc_slab * (hat.shape[2] // 16) * p.nc_slab * 16 * 8
    g.close()


def ck_closed_form(off, nc, comp):
    """coarse_kernel's real-space table on the periodic mesh (kernel_initialization.f90:293-336): -r_c / r^3 with r = 4 * (wrapped
    cell offset), replaced inside the 4^3 corner by wfxyzc.2.ascii with the sign of the mirrored component (:366-457)."""
    off = np.asarray(off, np.int64)
    w = np.where(off < nc // 2 + 1, off, off - nc).astype(np.float64)
    xs = 4.0 * w
    rr = np.sqrt((xs ** 2).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        v = np.where(rr == 0, 0.0, -xs[..., comp] / rr ** 3)
    t = np.where(off < 4, off, np.where(off > nc - 4, nc - off, -1))
    inside = (t >= 0).all(-1)
    sgn = np.where(off[..., comp] > nc - 4, -1.0, 1.0)
    tab = COARSE_TABLE[np.clip(t[..., 2], 0, 3), np.clip(t[..., 1], 0, 3), np.clip(t[..., 0], 0, 3), comp]
    return np.where(inside, sgn * tab, v)


@pytest.mark.parametrize("tiles", [1, 4], ids=["nc256", "nc1024"])
def test_coarse_force_of_point_masses_is_the_real_space_kernel(tiles):
    """F_c = ck (*) rho exactly (K_c = Im FFT(ck), ck odd in c): a unit mass at the origin returns the kernel table itself --
    the corner from wfxyzc.2.ascii, -r_c/r^3 beyond -- and a mass elsewhere the same field translated (rank to rank through the
    slab -> cube exchange and the halo)."""
    p, g = make_group(tiles)
    nc, n = p.nc_dim, p.nc_node_dim
    g.set_kernel_tables(None, COARSE_TABLE)
    zero = np.zeros((n, n, n), np.float32)

    def run(rank, cell, mass):
        for i in range(8):
            cube = zero
            if g.local_ranks[i] == rank:
                cube = zero.copy()
                cube[cell[2], cell[1], cell[0]] = mass
            g.set_coarse_density(i, cube)
        g.coarse_transform("force")

    run(0, (0, 0, 0), 1.0)
    f0 = g.coarse_force(0)                                      # [k][j][i][3] incl. the one-cell halo: cell (i,j,k) at index +1
    m = 48
    kk, jj, ii = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    off = np.stack([ii, jj, kk], -1)
    fmax = float(np.abs(COARSE_TABLE).max())
    for comp in range(3):
        want = ck_closed_form(off, nc, comp)
        got = f0[1:1 + m, 1:1 + m, 1:1 + m, comp]
        assert np.abs(got - want).max() < 3e-6 * fmax, comp
    # the low halo of rank 0 holds the cells at offset -1 (periodic: the far side of the neighbour rank)
    k2, j2 = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    want = ck_closed_form(np.stack([np.full((m, m), nc - 1), j2, k2], -1), nc, 0)
    assert np.abs(f0[1:1 + m, 1:1 + m, 0, 0] - want).max() < 3e-6 * fmax
    # translation: the same mass at cell (5, 7, 11) of rank 7 -> rank 7 sees rank 0's field moved by (5, 7, 11)
    run(7, (5, 7, 11), 1.0)
    f7 = g.coarse_force(7)
    a = f7[1 + 11:n + 1, 1 + 7:n + 1, 1 + 5:n + 1]
    b = f0[1:n + 1 - 11, 1:n + 1 - 7, 1:n + 1 - 5]
    assert np.abs(a - b).max() < 3e-6 * fmax
    # superposition with another mass on another rank: linear in rho
    for i in range(8):
        cube = zero.copy()
        if g.local_ranks[i] == 7:
            cube[11, 7, 5] = 1.0
        if g.local_ranks[i] == 0:
            cube[0, 0, 0] = 2.5
        g.set_coarse_density(i, cube)
    g.coarse_transform("force")
    f0b = g.coarse_force(0)
    run(7, (5, 7, 11), 1.0)
    f0c = g.coarse_force(0)                                     # rank 0's share of the field of the mass on rank 7
    assert np.abs(f0b - (2.5 * f0 + f0c)).max() < 6e-6 * fmax
    # repetition (the persistent x passes reuse their row tables across trips): the same transform of a random density
    # three times, every rank's force array bit-identical each time
    rng = np.random.default_rng(5)
    cubes = [rng.poisson(8.0, (n, n, n)).astype(np.float32) * 8.0 for _ in range(2)]
    ref = None
    for rep in range(3):
        for i in range(8):
            g.set_coarse_density(i, cubes[i & 1])
        g.coarse_transform("force")
        got = [g.coarse_force(i) for i in (0, 3, 7)]
        if ref is None:
            ref = got
        else:
            assert all(np.array_equal(a, b) for a, b in zip(got, ref)), rep
    g.close()


def test_coarse_only_group_refuses_the_particle_path():
    from cubep3m_amd.lib import P3MError

    p, g = make_group(1)
    with pytest.raises(P3MError):
        g.particle_mesh(0.5, 0.05, 0.05, 8.0)
    with pytest.raises(P3MError):
        g.upload_particles(0, np.zeros((10, 6), np.float32))
    # every other entry point that would touch records, cells or fine arrays: an error code, not a launch on null pointers
    for call in (lambda: g.download_particles(0), lambda: g.update_position(0.05, 0.05), lambda: g.projection(8.0),
                 lambda: g.coarse(8.0, 0)):
        with pytest.raises(P3MError):
            call()
    import ctypes as C

    ctx = g.L.p3m_hip_group_ctx(g.h, 0)      # the phase-level API reached through the rank's context refuses as well
    buf = np.zeros(64, np.float32)
    assert g.L.p3m_hip_update_position(ctx, 0.05, 0.05, None) == P3M_ESTATE
    assert g.L.p3m_hip_link_list_and_pass(ctx) == P3M_ESTATE
    assert g.L.p3m_hip_probe_coarse(ctx, 8.0, buf.ctypes.data_as(C.c_void_p), None) == P3M_ESTATE
    assert g.L.p3m_hip_delete_particles(ctx, None) == P3M_ESTATE
    assert g.L.p3m_hip_fine_mesh(ctx, 0.5, 0.05, 8.0) == P3M_ESTATE
    g.close()
