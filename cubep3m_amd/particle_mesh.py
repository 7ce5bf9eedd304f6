"""Host-side mirror of the reference's `particle_mesh` interface over the C ABI.

The reference entry point takes no arguments and works on COMMON-block state
(source_threads/particle_mesh_threaded.f90:2, cubep3m.fh:147-171); here that state is the
`ParticleMesh` object: parameters of `parameters`/`cubepm.par`, the Green's functions
(`fine_kernel`/`coarse_kernel`), the particle store `xv`/`PID`/`np_local`, and after each call
the four time-step limits `dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc` that `timestep`
(timestep.f90:106-115) consumes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .kernels import default_tables
from .params import P3MStepOut, Params


def _vec3(v):
    if v is None:
        return None, None
    a = np.ascontiguousarray(v, np.float32).reshape(3)
    return a.ctypes.data_as(C.c_void_p), a


class ParticleMesh:
    def __init__(self, params: Params, fine_table=None, coarse_table=None, set_kernels=True):
        self.params = params
        self.L = _lib.load()
        self._cp = params.to_c()
        h = C.c_void_p()
        _lib.check(self.L.p3m_hip_create(C.byref(self._cp), C.byref(h)))
        self.h = h
        self.last = None
        if set_kernels:
            if fine_table is None or coarse_table is None:
                fine_table, coarse_table = default_tables()
            self.set_kernel_tables(fine_table, coarse_table)

    @classmethod
    def from_handle(cls, handle, params: Params):
        """Wrap a context owned by someone else (a rank of a ParticleMeshGroup); close() is a no-op."""
        self = cls.__new__(cls)
        self.params, self.L, self.h, self.last, self._borrowed = params, _lib.load(), C.c_void_p(handle), None, True
        return self

    # -- lifecycle ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
            self.L.p3m_hip_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def derived(self, what):
        return int(self.L.p3m_hip_derived(self.h, what))

    # -- fine_kernel / coarse_kernel (kernel_initialization.f90) --------------------------
    def set_kernel_tables(self, fine_table, coarse_table):
        _lib.check(self.L.p3m_hip_set_kernel_tables(self.h, np.ascontiguousarray(fine_table, np.float32),
                                                    np.ascontiguousarray(coarse_table, np.float32)))

    def set_kernels_raw(self, kern_f, kern_c):
        _lib.check(self.L.p3m_hip_set_kernels_raw(self.h, np.ascontiguousarray(kern_f, np.float32), np.ascontiguousarray(kern_c, np.float32)))

    def get_kernels(self):
        p = self.params
        kf = np.empty((p.nf_tile, p.nf_tile, p.nf_tile // 2 + 1, 3), np.float32)
        kc = np.empty((p.nc_dim, p.nc_dim, p.nc_dim // 2 + 1, 3), np.float32)
        _lib.check(self.L.p3m_hip_get_kernels(self.h, kf.ctypes.data_as(C.c_void_p), kc.ctypes.data_as(C.c_void_p)))
        return kf, kc

    # -- xv / PID / np_local --------------------------------------------------------------
    def upload_particles(self, xv, pid=None):
        xv = np.ascontiguousarray(xv, np.float32).reshape(-1, 6)
        pp = None
        if pid is not None:
            pid = np.ascontiguousarray(pid, np.int64)
            pp = pid.ctypes.data_as(C.c_void_p)
        _lib.check(self.L.p3m_hip_upload_particles(self.h, xv.ctypes.data_as(C.c_void_p), pp, len(xv)))

    @property
    def np_local(self):
        n = C.c_int32()
        _lib.check(self.L.p3m_hip_download_particles(self.h, None, None, C.byref(n)))
        return n.value

    def download_particles(self):
        n = self.np_local
        xv = np.empty((n, 6), np.float32)
        pid = np.empty(n, np.int64)
        nn = C.c_int32()
        _lib.check(self.L.p3m_hip_download_particles(self.h, xv.ctypes.data_as(C.c_void_p), pid.ctypes.data_as(C.c_void_p), C.byref(nn)))
        return xv, pid

    # -- subroutine particle_mesh ----------------------------------------------------------
    def particle_mesh(self, a_mid, dt, dt_old, mass_p, offset=None, move_back=None) -> P3MStepOut:
        o = P3MStepOut()
        po, _a = _vec3(offset)
        pm, _b = _vec3(move_back)
        _lib.check(self.L.p3m_hip_particle_mesh(self.h, a_mid, dt, dt_old, mass_p, po, pm, C.byref(o)))
        self.last = o
        return o

    PHASES = ("update_position", "link_list", "particle_pass", "fine_mass", "fine_fft", "fine_kick", "pp_intra", "pp_ext",
              "coarse_mass", "coarse_force", "coarse_velocity", "delete_particles")

    def phase_timing(self, on=True):
        """per-phase GPU times of the following steps (timers.f90:68-77, -DMPI_TIME)"""
        _lib.check(self.L.p3m_hip_phase_timing(self.h, 1 if on else 0))

    def last_phase_ms(self):
        ms = np.zeros(12, np.float32)
        _lib.check(self.L.p3m_hip_last_phase_ms(self.h, ms))
        return dict(zip(self.PHASES, (float(v) for v in ms)))

    # -- phases, in the order particle_mesh calls them ---------------------------------------
    def update_position(self, dt, dt_old, offset=None):
        po, _a = _vec3(offset)
        _lib.check(self.L.p3m_hip_update_position(self.h, dt, dt_old, po))

    def link_list_and_pass(self):
        _lib.check(self.L.p3m_hip_link_list_and_pass(self.h))

    def fine_mesh(self, a_mid, dt, mass_p):
        _lib.check(self.L.p3m_hip_fine_mesh(self.h, a_mid, dt, mass_p))

    def coarse_mesh(self, a_mid, dt, mass_p):
        _lib.check(self.L.p3m_hip_coarse_mesh(self.h, a_mid, dt, mass_p))

    def delete_particles(self, move_back=None):
        pm, _b = _vec3(move_back)
        _lib.check(self.L.p3m_hip_delete_particles(self.h, pm))

    def step_out(self, a_mid):
        o = P3MStepOut()
        _lib.check(self.L.p3m_hip_get_step_out(self.h, a_mid, C.byref(o)))
        return o

    # -- probes ------------------------------------------------------------------------------
    def tile_density(self, tile, mass_p):
        nf = self.params.nf_tile
        rho = np.empty((nf, nf, nf + 2), np.float32)
        _lib.check(self.L.p3m_hip_probe_tile_density(self.h, tile[0], tile[1], tile[2], mass_p, rho))
        return rho

    def tile_force(self, rho):
        pt = self.params.nf_physical_tile_dim
        f = np.empty((pt + 3, pt + 3, pt + 3, 3), np.float32)
        m = C.c_float()
        _lib.check(self.L.p3m_hip_probe_tile_force(self.h, np.ascontiguousarray(rho, np.float32), f, C.byref(m)))
        return f, m.value

    def projection(self, mass_p):
        """projection.f90 on the sorted records with ghosts (call link_list_and_pass before, delete_particles after):
        (pxy[y][x], pxz[z][x], pyz[z][y], projected mass) of this rank."""
        n = self.params.nf_physical_node_dim * self.params.nodes_dim
        maps = [np.empty((n, n), np.float32) for _ in range(3)]
        tot = C.c_double()
        _lib.check(self.L.p3m_hip_projection(self.h, mass_p, *(m.ctypes.data_as(C.c_void_p) for m in maps), C.byref(tot)))
        return maps[0], maps[1], maps[2], tot.value

    def coarse_power(self, mass_p, box):
        """coarse_power.f90 on the coarse density of the last particle_mesh step: (nc_dim, 2) rows (k, Delta^2(k))."""
        ps = np.zeros((self.params.nc_dim, 2), np.float32)
        _lib.check(self.L.p3m_hip_coarse_power(self.h, mass_p, box, ps))
        return ps

    def coarse(self, mass_p, want_force=True):
        p = self.params
        rho = np.empty((p.nc_node_dim,) * 3, np.float32)
        f = np.empty((p.nc_node_dim + 2,) * 3 + (3,), np.float32) if want_force else None
        _lib.check(self.L.p3m_hip_probe_coarse(self.h, mass_p, rho.ctypes.data_as(C.c_void_p),
                                               f.ctypes.data_as(C.c_void_p) if want_force else None))
        return rho, f

    def fft3d(self, a, n, direction):
        a = np.ascontiguousarray(a, np.float32).copy()
        _lib.check(self.L.p3m_hip_fft3d(self.h, a, n, direction))
        return a

    def time_fine_sweep(self, mass_p, reps=5):
        ms = C.c_float()
        _lib.check(self.L.p3m_hip_time_fine_sweep(self.h, mass_p, reps, C.byref(ms)))
        return ms.value

    def time_fine_gather(self, reps=5):
        """Average ms of the gather half of the fine mesh (maximum + interpolation + kick, dt = 0) over the boxes of the last sweep."""
        ms = C.c_float()
        _lib.check(self.L.p3m_hip_time_fine_gather(self.h, reps, C.byref(ms)))
        return ms.value

    def time_pp(self, a_mid, dt, mass_p, reps=5):
        """(ms per k_pp_intra launch, ms per extended-PP launch, pair evaluations of each) on the sorted records with
        ghosts (after link_list_and_pass)."""
        a, b, na, nb = C.c_float(), C.c_float(), C.c_int64(), C.c_int64()
        _lib.check(self.L.p3m_hip_time_pp(self.h, a_mid, dt, mass_p, reps, C.byref(a), C.byref(b), C.byref(na), C.byref(nb)))
        return a.value, b.value, na.value, nb.value

    FFT_PASSES = ("x_fwd", "y_fwd", "z_fwd", "z_inv_fused", "y_inv", "x_inv_extract", "z_inv_multiply")   # 7: "x_inv_kick_fused" (after a whole NGP step)

    def time_fft_pass(self, which, reps=20):
        """Average ms per launch of one FFT pass kernel over the tile batch (HIP events on the library stream)."""
        ms, nb = C.c_float(), C.c_int32()
        _lib.check(self.L.p3m_hip_time_fft_pass(self.h, which, reps, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value
