"""coarse_power.f90 (second half): the mass power spectrum of the coarse density.
FFT-dependent, so the reference's object code cannot be run here (FFTW 2.1.5 absent): the oracle's restatement is pinned
by closed forms (a single plane wave, with the reference's binning, its sinc deconvolution of the imaginary part only and
its counting of the kx = 0 plane rebuilt independently in numpy), the HIP path is held to the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
from common import COARSE_TABLE, FINE_TABLE, cfg1, clustered_particles

BOX = 200.0


def bin_counts(nc):
    """modes per bin k1 = ceiling(|k|) of the half spectrum as coarse_power.f90:45-98 walks it"""
    kx = np.arange(nc // 2 + 1)[None, None, :]
    kf = np.where(np.arange(nc) < nc // 2 + 1, np.arange(nc), np.arange(nc) - nc)
    ky, kz = kf[None, :, None], kf[:, None, None]
    kr = np.sqrt((kx ** 2 + ky ** 2 + kz ** 2).astype(np.float32))
    skip = ((kx == 0) & (ky <= 0) & (kz <= 0)) | ((kx == 0) & (ky > 0) & (kz < 0)) | (kr == 0)
    k1 = np.ceil(kr).astype(int)
    return np.bincount(k1[~skip].ravel(), minlength=nc + 2)


@pytest.mark.parametrize("phase,axis", [("cos", 0), ("sin", 0), ("cos", 2), ("sin", 1)])
def test_oracle_power_of_a_plane_wave(phase, axis):
    p = cfg1()                                   # nc_dim = 16
    o = ol.Oracle(p)
    nc, m, amp, mass_p = p.nc_dim, 3, 0.25, 8.0
    mean = (p.nf_physical_node_dim / 2) ** 3 * mass_p / nc ** 3        # coarse_power.f90:24
    x = np.arange(nc, dtype=np.float64)
    w = np.cos(2 * np.pi * m * x / nc) if phase == "cos" else np.sin(2 * np.pi * m * x / nc)
    shape = [1, 1, 1]
    shape[2 - axis] = nc                         # arrays are [z][y][x]
    o.rho_c_view(0)[...] = (mean * (1.0 + amp * w.reshape(shape))).astype(np.float32) * np.ones((nc, nc, nc), np.float32)
    ps = o.coarse_power(mass_p, BOX)
    cnt = bin_counts(nc)
    # delta-hat(m) / N^3 = amp/2 (cos: real) or -i amp/2 (sin: imaginary, divided by sinc^4 as :96 does); the half spectrum
    # holds +m only along x, and on the kx = 0 plane :60-61 keep one member of each conjugate pair: one mode either way
    sinc = np.sinc(m / nc)                       # sin(pi m / nc) / (pi m / nc)
    pw = (amp / 2) ** 2 / (sinc ** 4 if phase == "sin" else 1.0)
    nmodes = 1
    want = 4 * np.float32(3.141592654) * (m - 1) ** 3 * (nmodes * pw) / cnt[m]
    assert ps[m - 1, 1] == pytest.approx(want, rel=2e-5)               # bin k1 = m is row m (1-based)
    assert ps[m - 1, 0] == pytest.approx(2 * 3.141592654 * (m - 1) / BOX, rel=1e-6)
    others = np.delete(ps[:, 1], m - 1)
    assert np.abs(others).max() < 1e-6 * want
    # k column: 2 pi (bin-1)/box wherever the bin holds modes, the raw zero weight elsewhere
    for k in range(1, nc + 1):
        if cnt[k]:
            assert ps[k - 1, 0] == pytest.approx(2 * 3.141592654 * (k - 1) / BOX, rel=1e-6)
        else:
            assert ps[k - 1, 0] == 0 and ps[k - 1, 1] == 0


def test_power_file_format(tmp_path):
    from cubep3m_amd import io_formats

    ps = np.array([[0.0, 0.0], [0.0314159265, 1.25e-3], [12.5, 123456.789]], np.float32)
    f = tmp_path / "0.000ps.dat"
    io_formats.write_power(f, ps)
    lines = open(f).read().splitlines()
    assert len(lines) == 3 and all(len(l) == 40 for l in lines)       # '(2f20.10)'
    got = np.array([[float(l[:20]), float(l[20:])] for l in lines])
    assert np.allclose(got, ps.astype(np.float64), rtol=0, atol=6e-11)
    assert lines[1][:20] == "%20.10f" % float(ps[1, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("nd,pencil", [(1, False), (2, False), (2, True)])
def test_hip_coarse_power_vs_oracle(nd, pencil):
    from cubep3m_amd.group import ParticleMeshGroup

    p = cfg1(nodes_dim=nd, pencil=pencil)
    box = 64.0 * nd
    xv = clustered_particles(40000 * nd ** 3, box, seed=17, frac=0.4, nblobs=30, sigma=1.5)
    g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE)
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    parts = g.scatter_global(xv, np.arange(1, len(xv) + 1, dtype=np.int64))
    for r, (a, b) in parts.items():
        o.set_particles(r, a, b)
    g.particle_mesh(0.5, 0.0, 0.0, 8.0)          # dt = 0: nothing moves; leaves the step's rho-hat on the device
    o.link_list()
    assert o.particle_pass() == 0
    o.coarse_density(8.0)
    want = o.coarse_power(8.0, BOX)
    got = g.coarse_power(8.0, BOX)
    assert np.array_equal(got[:, 0], want[:, 0])                       # k values and empty bins
    m = want[:, 1] != 0
    assert m.sum() > p.nc_dim // 2
    assert np.abs(got[m, 1] / want[m, 1] - 1).max() < 2e-4            # the oracle adds in real(4) like the reference, the device in double
