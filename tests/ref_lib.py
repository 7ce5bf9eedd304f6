"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/<cfg>/libref.so -- the reference's own
FFT-free object code (built where it lies by oracle/build_ref.sh) behind oracle/ref_driver.f90."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def so_path(cfg):
    return os.path.join(ROOT, "oracle", "_ref", cfg, "libref.so")


def available(cfg):
    return os.path.exists(so_path(cfg))


class Ref:
    """One MPI rank of the reference (COMMON blocks are process-global: one Ref per process)."""

    def __init__(self, cfg):
        L = C.CDLL(so_path(cfg))
        L.ref_sizes.argtypes = [i32p]
        L.ref_neighbors.argtypes = [i32p]
        L.ref_set_scalars.argtypes = [C.c_float] * 4
        L.ref_set_particles.argtypes = [f32p, i64p, C.c_int]
        L.ref_np_local.restype = C.c_int
        L.ref_get_particles.argtypes = [f32p, i64p]
        L.ref_get_lists.argtypes = [i32p, i32p]
        L.ref_fine_deposit.argtypes = [i32p, C.c_int, f32p]
        L.ref_coarse_mass.argtypes = [f32p]
        if hasattr(L, "ref_fine_velocity"):
            L.ref_fine_velocity.argtypes = [i32p, f32p, f32p]
        L.ref_set_force_c.argtypes = [f32p]
        L.ref_get_force_c.argtypes = [f32p]
        L.ref_coarse_max_dt.restype = C.c_float
        self.L = L
        L.ref_init()
        s = np.zeros(16, np.int32)
        L.ref_sizes(s)
        (self.nodes_dim, self.tiles_node_dim, self.nf_tile, self.nf_buf, self.nc_node_dim, self.nc_dim,
         self.max_np, self.hoc_nc_l, self.hoc_nc_h, self.nf_physical_node_dim, self.rank, self.cores,
         self.max_buf) = (int(v) for v in s[:13])
        self.cart_coords = [int(v) for v in s[13:16]]

    def neighbors(self):
        n = np.zeros(6, np.int32)
        self.L.ref_neighbors(n)
        return n

    def set_scalars(self, a_mid, dt, dt_old, mass_p):
        self.L.ref_set_scalars(a_mid, dt, dt_old, mass_p)

    def set_particles(self, xv, pid):
        self.L.ref_set_particles(np.ascontiguousarray(xv, np.float32), np.ascontiguousarray(pid, np.int64), len(xv))

    def get_particles(self):
        n = self.L.ref_np_local()
        xv = np.empty((n, 6), np.float32)
        pid = np.empty(n, np.int64)
        self.L.ref_get_particles(xv, pid)
        return xv, pid

    def get_lists(self):
        hn = self.hoc_nc_h - self.hoc_nc_l + 1
        hoc = np.empty(hn ** 3, np.int32)
        ll = np.empty(max(self.L.ref_np_local(), 1), np.int32)
        self.L.ref_get_lists(hoc, ll)
        return hoc.reshape(hn, hn, hn), ll

    def update_position(self):
        self.L.ref_update_position()

    def link_list(self):
        self.L.ref_link_list()

    def particle_pass(self):
        self.L.ref_particle_pass()

    def delete_particles(self):
        self.L.ref_delete_particles()

    def fine_deposit(self, tile, ngp):
        nf = self.nf_tile
        rho = np.empty((nf, nf, nf + 2), np.float32)
        self.L.ref_fine_deposit(np.asarray(tile, np.int32), 1 if ngp else 0, rho)
        return rho

    def fine_velocity(self, tile, f):
        """f: force box [k][j][i][3] of (pt+3)^3 cells -> (max |F|, pp_force_max) after fine_velocity.f90 ran on it."""
        out = np.zeros(2, np.float32)
        self.L.ref_fine_velocity(np.asarray(tile, np.int32), np.ascontiguousarray(f, np.float32), out)
        return float(out[0]), float(out[1])

    def coarse_mass(self):
        n = self.nc_node_dim
        rho = np.empty((n, n, n), np.float32)
        self.L.ref_coarse_mass(rho)
        return rho

    def set_force_c(self, f):  # f: [k][j][i][3] interior
        self.L.ref_set_force_c(np.ascontiguousarray(f, np.float32))

    def get_force_c(self):
        n = self.nc_node_dim + 2
        f = np.empty((n, n, n, 3), np.float32)
        self.L.ref_get_force_c(f)
        return f

    def coarse_force_buffer(self):
        self.L.ref_coarse_force_buffer()

    def coarse_max_dt(self):
        return float(self.L.ref_coarse_max_dt())

    def coarse_velocity(self):
        self.L.ref_coarse_velocity()
