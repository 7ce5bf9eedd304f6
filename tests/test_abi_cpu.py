"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/p3m_hip.h declares, mirrors the header's struct layouts, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from cubep3m_amd import lib
from cubep3m_amd.params import P3MParams, P3MStepOut, Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    if not os.path.exists(lib.SO_PATH):
        lib.build()
    return C.CDLL(lib.SO_PATH)


def test_every_declared_symbol_is_exported(so):
    hdr = open(os.path.join(ROOT, "include", "p3m_hip.h")).read()
    declared = set(re.findall(r"\b(p3m_hip_\w+|particle_mesh_hip_)\s*\(", hdr))
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for s in declared:
        assert hasattr(so, s), s


def test_struct_layouts_match_header():
    assert C.sizeof(P3MParams) == 15 * 4
    assert P3MStepOut.sum_rho_f.offset == 16 and P3MStepOut.np_total.offset == 32
    assert C.sizeof(P3MStepOut) == 72


def test_params_derived_sizes_follow_cubepm_par():
    p = Params(nodes_dim=2, tiles_node_dim=2, nf_tile=80)
    assert (p.nf_physical_tile_dim, p.nf_physical_node_dim, p.nc_node_dim, p.nc_dim, p.nc_slab) == (32, 64, 16, 32, 4)
    with pytest.raises(ValueError):
        Params(nf_tile=81).validate()
    with pytest.raises(ValueError):
        Params(ngp=False, ppint=True).validate()


def test_fails_loudly_without_gpu(so):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cubep3m_amd.particle_mesh import ParticleMesh

    with pytest.raises(lib.P3MError) as e:
        ParticleMesh(Params())
    assert "no HIP device" in str(e.value)
