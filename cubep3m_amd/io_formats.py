"""The reference's particle files through the C ABI (include/p3m_hip.h): `xv<rank>.ic`, `<z>xv<rank>.dat`,
`<z>PID<rank>.dat` (particle_initialization.f90:296-332, checkpoint.f90:22-124), in both layouts the reference can be
compiled for: form='unformatted' (default; one framed record per WRITE) and form='binary' (-DBINARY; a byte stream)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


class P3MCkptHeader(C.Structure):
    _fields_ = [("np_local", C.c_int32), ("a", C.c_float), ("t", C.c_float), ("tau", C.c_float), ("nts", C.c_int32), ("dt_f_acc", C.c_float),
                ("dt_pp_acc", C.c_float), ("dt_c_acc", C.c_float), ("cur_checkpoint", C.c_int32), ("cur_projection", C.c_int32),
                ("cur_halofind", C.c_int32), ("mass_p", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def checkpoint_names(z, rank):
    """File names of checkpoint.f90:27-37: z as f7.3, left-adjusted."""
    zs = ("%7.3f" % z).strip()
    return "%sxv%d.dat" % (zs, rank), "%sPID%d.dat" % (zs, rank)


def projection_names(z):
    """File names of projection.f90:59-87: z as f7.3, left-adjusted, then proj_xy / proj_xz / proj_yz."""
    zs = ("%7.3f" % z).strip()
    return tuple("%sproj_%s.dat" % (zs, ax) for ax in ("xy", "xz", "yz"))


def _b(path):
    return str(path).encode()


def write_projection(path, a, m, binary=False):
    """One projection file (projection.f90:104-113): the scale factor, then the square map in the reference's order."""
    m = np.ascontiguousarray(m, np.float32)
    assert m.ndim == 2 and m.shape[0] == m.shape[1]
    _lib.check(_lib.load().p3m_hip_write_projection(_b(path), float(a), m.ctypes.data_as(C.c_void_p), m.shape[0], int(binary)))


def read_projection(path, n, binary=False):
    a = C.c_float()
    m = np.empty((n, n), np.float32)
    _lib.check(_lib.load().p3m_hip_read_projection(_b(path), C.byref(a), m.ctypes.data_as(C.c_void_p), n, int(binary)))
    return a.value, m


def write_checkpoint(path, header: P3MCkptHeader, xv, shake_offset=None, binary=False, ppint=False):
    xv = np.ascontiguousarray(xv, np.float32).reshape(-1, 6)
    header.np_local = len(xv)
    so = None if shake_offset is None else np.ascontiguousarray(shake_offset, np.float32)
    _lib.check(_lib.load().p3m_hip_write_checkpoint(_b(path), C.byref(header), xv.ctypes.data_as(C.c_void_p),
                                                    None if so is None else so.ctypes.data_as(C.c_void_p), int(binary), int(ppint)))


def read_checkpoint(path, binary=False, ppint=False):
    L = _lib.load()
    h = P3MCkptHeader()
    _lib.check(L.p3m_hip_read_checkpoint(_b(path), C.byref(h), None, 0, int(binary), int(ppint)))
    xv = np.empty((h.np_local, 6), np.float32)
    _lib.check(L.p3m_hip_read_checkpoint(_b(path), C.byref(h), xv.ctypes.data_as(C.c_void_p), len(xv), int(binary), int(ppint)))
    return h, xv


def write_pid_checkpoint(path, header: P3MCkptHeader, pid, binary=False, ppint=False):
    pid = np.ascontiguousarray(pid, np.int64)
    header.np_local = len(pid)
    _lib.check(_lib.load().p3m_hip_write_pid_checkpoint(_b(path), C.byref(header), pid.ctypes.data_as(C.c_void_p), int(binary), int(ppint)))


def read_pid_checkpoint(path, binary=False, ppint=False):
    L = _lib.load()
    h = P3MCkptHeader()
    _lib.check(L.p3m_hip_read_pid_checkpoint(_b(path), C.byref(h), None, 0, int(binary), int(ppint)))
    pid = np.empty(h.np_local, np.int64)
    _lib.check(L.p3m_hip_read_pid_checkpoint(_b(path), C.byref(h), pid.ctypes.data_as(C.c_void_p), len(pid), int(binary), int(ppint)))
    return h, pid


def write_ic(path, xv, binary=False):
    xv = np.ascontiguousarray(xv, np.float32).reshape(-1, 6)
    _lib.check(_lib.load().p3m_hip_write_ic(_b(path), xv.ctypes.data_as(C.c_void_p), len(xv), int(binary)))


def read_ic(path, binary=False, max_np=None):
    L = _lib.load()
    n = C.c_int32()
    _lib.check(L.p3m_hip_read_ic(_b(path), None, 0, C.byref(n), int(binary)))
    cap = n.value if max_np is None else max_np
    xv = np.empty((min(n.value, cap) if max_np is None else n.value, 6), np.float32)
    _lib.check(L.p3m_hip_read_ic(_b(path), xv.ctypes.data_as(C.c_void_p), cap, C.byref(n), int(binary)))
    return xv


def write_power(path, ps):
    """<z>ps.dat of coarse_power.f90:121-133: one formatted '(2f20.10)' line per bin."""
    ps = np.ascontiguousarray(ps, np.float32)
    _lib.check(_lib.load().p3m_hip_write_power(_b(path), ps, ps.shape[0]))
