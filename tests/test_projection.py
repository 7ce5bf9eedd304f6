"""Density projections (projection.f90).  CPU part: the oracle restatement and the file writer
against what the REFERENCE'S OWN projection.o computed and wrote (tests/golden/ref_projection.npz, made by
tests/golden/make_ref_projection.py).  GPU part: the HIP path against the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from common import cfg1, clustered_particles, rel_rms
from cubep3m_amd import io_formats as iof
from cubep3m_amd.params import Params

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_projection.npz"))


def oracle_maps(p, xv, pid, mass_p):
    o = ol.Oracle(p)
    o.set_particles(0, xv, pid)
    o.link_list()
    o.particle_pass()
    return o.projection(mass_p)


def test_oracle_projection_is_the_reference_bit_for_bit():
    p = cfg1(nodes_dim=1)
    pxy, pxz, pyz, tot = oracle_maps(p, G["xv"], G["pid"], float(G["rv"][1]))
    assert np.array_equal(pxy, G["pxy"]) and np.array_equal(pxz, G["pxz"]) and np.array_equal(pyz, G["pyz"])
    assert tot == pytest.approx(len(G["xv"]) * float(G["rv"][1]), rel=1e-6)      # every particle's mass lands in the interior once
    assert float(pxy.sum()) == pytest.approx(tot, rel=1e-5)


def test_projection_files_are_the_bytes_the_reference_wrote(tmp_path):
    a, z = float(G["rv"][0]), float(G["rv"][2])
    names = iof.projection_names(z)
    assert list(names) == [str(G["name_" + ax]) for ax in ("xy", "xz", "yz")]
    for ax, name, m in zip(("xy", "xz", "yz"), names, (G["pxy"], G["pxz"], G["pyz"])):
        f = tmp_path / name
        iof.write_projection(f, a, m, binary=False)
        assert np.array_equal(np.fromfile(f, np.uint8), G["file_" + ax])
        a2, m2 = iof.read_projection(f, m.shape[0], binary=False)
        assert a2 == np.float32(a) and np.array_equal(m2, m)
        iof.write_projection(f, a, m, binary=True)                                  # -DBINARY: no record markers
        assert os.path.getsize(f) == 4 + m.size * 4
        a3, m3 = iof.read_projection(f, m.shape[0], binary=True)
        assert a3 == np.float32(a) and np.array_equal(m3, m)
    with pytest.raises(Exception):
        iof.read_projection(tmp_path / names[0], 32, binary=False)                  # wrong size: record length mismatch


@pytest.mark.gpu
@pytest.mark.parametrize("ngp", [True, False])
def test_hip_projection_vs_oracle(ngp):
    from cubep3m_amd.particle_mesh import ParticleMesh
    p = cfg1(nodes_dim=1, ngp=ngp)
    box = float(p.nf_physical_node_dim)
    n = int(box ** 3 / 8)
    xv = clustered_particles(n, box, seed=41, frac=0.3, nblobs=12, sigma=1.5, vel_sigma=0.3)
    pid = np.arange(1, n + 1, dtype=np.int64)
    want = oracle_maps(p, xv, pid, 8.0)
    g = ParticleMesh(p)
    g.upload_particles(xv, pid)
    g.link_list_and_pass()
    got = g.projection(8.0)
    g.delete_particles()
    for a, b in zip(got[:3], want[:3]):
        assert rel_rms(a, b) < 1e-6 and np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    assert got[3] == pytest.approx(want[3], rel=1e-6) and got[3] == pytest.approx(8.0 * n, rel=1e-6)
    xo, po = g.download_particles()                                                 # the ghosts are gone again, nothing moved
    o = np.argsort(po)
    assert len(po) == n and np.array_equal(xo[o], xv)


@pytest.mark.gpu
def test_hip_projection_on_eight_logical_ranks_vs_oracle():
    """Only the ranks at coordinate 0 of the projected axis contribute (projection.f90:170-181): a slab, not the volume."""
    from cubep3m_amd.group import ParticleMeshGroup
    p = cfg1(nodes_dim=2)
    box = float(p.nf_physical_node_dim)
    n = int(box ** 3 / 8)
    o = ol.Oracle(p)
    grp = ParticleMeshGroup(p)
    for i, r in enumerate(grp.local_ranks):
        xv = clustered_particles(n, box, seed=50 + r, frac=0.3, nblobs=8, sigma=1.5)
        pid = np.arange(1, n + 1, dtype=np.int64) + r * n
        o.set_particles(r, xv, pid)
        grp.upload_particles(i, xv, pid)
    o.link_list()
    o.particle_pass()
    want = o.projection(8.0)
    got = grp.projection(8.0)
    for a, b in zip(got[:3], want[:3]):
        assert rel_rms(a, b) < 1e-6
    assert got[3] == pytest.approx(want[3], rel=1e-6) and got[3] == pytest.approx(8.0 * n * 8, rel=1e-6)
    assert float(got[0].sum()) == pytest.approx(float(want[0].sum()), rel=1e-5) and float(got[0].sum()) < 0.6 * got[3]   # a slab: about half of the mass
    assert sum(grp.download_particles(i)[0].shape[0] for i in range(len(grp.local_ranks))) == 8 * n
    grp.close()
