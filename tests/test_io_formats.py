"""Particle files: `xv<rank>.ic`, `<z>xv<rank>.dat`, `<z>PID<rank>.dat`.
tests/golden/ref_io.npz holds what the REFERENCE'S OWN object code did (tests/golden/make_ref_io.py, oracle/_ref):
  * particle_initialization.o read an IC file written by cubep3m_amd.io_formats -> the particles it ended up with;
  * checkpoint.o wrote the checkpoint files of a given state (builds without and with -DPPINT) -> their bytes.
The C ABI (host code of libp3m_hip.so, no GPU needed) must write those bytes and read them back."""
import os

import numpy as np
import pytest

from cubep3m_amd import io_formats as iof
from cubep3m_amd.lib import P3MError

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_io.npz"))


def header_of(tag, n):
    rv, iv = G[tag + "_rv"], G[tag + "_iv"]
    h = iof.P3MCkptHeader()
    h.np_local = n
    h.a, h.t, h.tau, h.dt_f_acc, h.dt_pp_acc, h.dt_c_acc, h.mass_p = (float(rv[i]) for i in (0, 1, 2, 3, 4, 5, 6))
    h.nts, h.cur_projection, h.cur_halofind = int(iv[0]), int(iv[2]), int(iv[3])
    h.cur_checkpoint = int(iv[1]) + 1          # checkpoint.f90:52 increments before it writes the header
    return h, rv[8:11].copy()


@pytest.mark.parametrize("tag,ppint", [("pm", False), ("pp", True)])
def test_ic_file_is_read_by_the_reference_and_rewritten_identically(tag, ppint, tmp_path):
    assert np.array_equal(G[tag + "_ref_read_xv"], G[tag + "_ic_xv"])      # the reference read exactly what was written
    f = tmp_path / "xv0.ic"
    iof.write_ic(f, G[tag + "_ic_xv"])
    assert np.array_equal(np.fromfile(f, np.uint8), G[tag + "_ic_bytes"])
    assert np.array_equal(iof.read_ic(f), G[tag + "_ic_xv"])
    with pytest.raises(P3MError):                                          # "too many particles to store" (:317-321)
        iof.read_ic(f, max_np=10)


@pytest.mark.parametrize("tag,ppint", [("pm", False), ("pp", True)])
def test_checkpoint_files_are_the_references_bytes(tag, ppint, tmp_path):
    xv, pid = G[tag + "_ref_read_xv"], G[tag + "_ref_pid"]
    h, shake = header_of(tag, len(xv))
    name_xv, name_pid = iof.checkpoint_names(float(G[tag + "_rv"][7]), 0)
    assert name_xv == str(G[tag + "_ckpt_name"])                            # '2.200xv0.dat' (f7.3, adjustl)
    iof.write_checkpoint(tmp_path / name_xv, h, xv, shake_offset=shake, ppint=ppint)
    assert np.array_equal(np.fromfile(tmp_path / name_xv, np.uint8), G[tag + "_ckpt_xv_bytes"])
    iof.write_pid_checkpoint(tmp_path / name_pid, h, pid, ppint=ppint)
    assert np.array_equal(np.fromfile(tmp_path / name_pid, np.uint8), G[tag + "_ckpt_pid_bytes"])
    # and back (particle_initialization.f90:114-145 reads them with the same statements)
    G[tag + "_ckpt_xv_bytes"].tofile(tmp_path / "r.dat")
    hr, xr = iof.read_checkpoint(tmp_path / "r.dat", ppint=ppint)
    want = h.as_dict()
    if not ppint:
        want["dt_pp_acc"] = 0.0                                             # not in the file without -DPPINT
    assert hr.as_dict() == want
    exp = xv.copy()
    exp[:, :3] -= shake                                                     # checkpoint.f90:80
    assert np.array_equal(xr, exp)
    G[tag + "_ckpt_pid_bytes"].tofile(tmp_path / "p.dat")
    hp, pr = iof.read_pid_checkpoint(tmp_path / "p.dat", ppint=ppint)
    assert hp.as_dict() == want and np.array_equal(pr, pid)
    # the wrong -DPPINT setting is an error, not garbage
    with pytest.raises(P3MError):
        iof.read_checkpoint(tmp_path / "r.dat", ppint=not ppint)


def test_binary_layout_is_the_bare_payload(tmp_path):
    xv = G["pm_ic_xv"]
    h, shake = header_of("pp", len(xv))
    iof.write_ic(tmp_path / "b.ic", xv, binary=True)
    raw = np.fromfile(tmp_path / "b.ic", np.uint8)
    assert len(raw) == 4 + 24 * len(xv) and raw[:4].view(np.int32)[0] == len(xv) and np.array_equal(raw[4:].view(np.float32).reshape(-1, 6), xv)
    assert np.array_equal(iof.read_ic(tmp_path / "b.ic", binary=True), xv)
    iof.write_checkpoint(tmp_path / "b.dat", h, xv, shake_offset=shake, binary=True, ppint=True)
    assert os.path.getsize(tmp_path / "b.dat") == 48 + 24 * len(xv)
    hr, xr = iof.read_checkpoint(tmp_path / "b.dat", binary=True, ppint=True)
    exp = xv.copy()
    exp[:, :3] -= shake
    assert hr.as_dict() == h.as_dict() and np.array_equal(xr, exp)
    # the unformatted file of the same data is the same payload with record markers around every WRITE
    iof.write_checkpoint(tmp_path / "u.dat", h, xv, shake_offset=shake, binary=False, ppint=True)
    assert os.path.getsize(tmp_path / "u.dat") == (48 + 8) + (24 + 8) * len(xv)
