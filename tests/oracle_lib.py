"""TEST INFRASTRUCTURE: ctypes binding of oracle/_build/libp3m_oracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from cubep3m_amd.params import P3MParams, P3MStepOut, Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libp3m_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build():
    src = os.path.join(ROOT, "oracle", "p3m_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(P3MParams)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_derived.restype = C.c_int64
        L.orc_derived.argtypes = [C.c_void_p, C.c_int]
        L.orc_fine_kernel.argtypes = [C.c_void_p, f32p]
        L.orc_coarse_kernel.argtypes = [C.c_void_p, f32p]
        L.orc_kern_f.restype = C.POINTER(C.c_float)
        L.orc_kern_f.argtypes = [C.c_void_p]
        L.orc_kern_c.restype = C.POINTER(C.c_float)
        L.orc_kern_c.argtypes = [C.c_void_p]
        L.orc_set_particles.argtypes = [C.c_void_p, C.c_int, f32p, C.c_void_p, C.c_int]
        L.orc_get_np.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_particles.argtypes = [C.c_void_p, C.c_int, f32p, i64p]
        L.orc_update_position.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_link_list.argtypes = [C.c_void_p]
        L.orc_particle_pass.argtypes = [C.c_void_p]
        L.orc_fine_mesh.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_coarse_mesh.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_move_grid_back.argtypes = [C.c_void_p, f32p]
        L.orc_delete_particles.argtypes = [C.c_void_p]
        L.orc_step_out.argtypes = [C.c_void_p, C.c_float, C.POINTER(P3MStepOut)]
        L.orc_particle_mesh.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(P3MStepOut)]
        L.orc_tile_density.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, f32p]
        L.orc_tile_force.argtypes = [C.c_void_p, f32p, f32p, C.POINTER(C.c_float)]
        L.orc_tile_velocity.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_float, C.c_float, C.c_float, f32p]
        L.orc_projection.argtypes = [C.c_void_p, C.c_float, f32p, f32p, f32p, C.POINTER(C.c_double)]
        L.orc_coarse_density.argtypes = [C.c_void_p, C.c_float]
        L.orc_rho_c.restype = C.POINTER(C.c_float)
        L.orc_rho_c.argtypes = [C.c_void_p, C.c_int]
        L.orc_force_c.restype = C.POINTER(C.c_float)
        L.orc_force_c.argtypes = [C.c_void_p, C.c_int]
        L.orc_coarse_force.argtypes = [C.c_void_p]
        L.orc_coarse_power.argtypes = [C.c_void_p, C.c_float, C.c_float, f32p]
        L.orc_distribute_force.argtypes = [C.c_void_p, f32p]
        L.orc_coarse_max_dt_and_velocity.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.orc_fft3d.argtypes = [f32p, C.c_int, C.c_int]
        L.orc_fft3d_rect.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _vec3(v):
    if v is None:
        return None
    a = np.ascontiguousarray(v, np.float32)
    return a.ctypes.data_as(C.c_void_p), a


class _RawOracle:
    """All nodes_dim^3 ranks of the reference simulated in one process."""

    def __init__(self, params: Params):
        self.p = params
        self.L = lib()
        self._cp = params.to_c()
        self.h = self.L.orc_create(C.byref(self._cp))
        assert self.h

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # kernels
    def set_kernel_tables(self, fine, coarse):
        self.L.orc_fine_kernel(self.h, np.ascontiguousarray(fine, np.float32))
        self.L.orc_coarse_kernel(self.h, np.ascontiguousarray(coarse, np.float32))

    def kern_f(self):
        nf = self.p.nf_tile
        return np.ctypeslib.as_array(self.L.orc_kern_f(self.h), shape=(nf, nf, nf // 2 + 1, 3)).copy()

    def kern_c(self):
        nc = self.p.nc_dim
        return np.ctypeslib.as_array(self.L.orc_kern_c(self.h), shape=(nc, nc, nc // 2 + 1, 3)).copy()

    # particles
    def set_particles(self, rank, xv, pid=None):
        xv = np.ascontiguousarray(xv, np.float32)
        pp = None if pid is None else np.ascontiguousarray(pid, np.int64).ctypes.data_as(C.c_void_p)
        r = self.L.orc_set_particles(self.h, rank, xv, pp, len(xv))
        assert r == 0, "oracle capacity (max_np) exceeded"

    def get_particles(self, rank=0):
        n = self.L.orc_get_np(self.h, rank)
        xv = np.empty((n, 6), np.float32)
        pid = np.empty(n, np.int64)
        self.L.orc_get_particles(self.h, rank, xv, pid)
        return xv, pid

    # phases
    def update_position(self, dt, dt_old, offset=None):
        o = _vec3(offset)
        self.L.orc_update_position(self.h, dt, dt_old, o[0] if o else None)

    def link_list(self):
        self.L.orc_link_list(self.h)

    def particle_pass(self):
        return self.L.orc_particle_pass(self.h)

    def fine_mesh(self, a_mid, dt, mass_p):
        self.L.orc_fine_mesh(self.h, a_mid, dt, mass_p)

    def coarse_mesh(self, a_mid, dt, mass_p):
        self.L.orc_coarse_mesh(self.h, a_mid, dt, mass_p)

    def delete_particles(self):
        self.L.orc_delete_particles(self.h)

    def step_out(self, a_mid=1.0):
        o = P3MStepOut()
        self.L.orc_step_out(self.h, a_mid, C.byref(o))
        return o

    def particle_mesh(self, a_mid, dt, dt_old, mass_p, offset=None, move_back=None):
        o = P3MStepOut()
        of, mb = _vec3(offset), _vec3(move_back)
        r = self.L.orc_particle_mesh(self.h, a_mid, dt, dt_old, mass_p, of[0] if of else None, mb[0] if mb else None, C.byref(o))
        assert r == 0, f"oracle particle_mesh failed {r}"
        return o

    # probes
    def tile_density(self, rank, tile, mass_p):
        nf = self.p.nf_tile
        rho = np.empty((nf, nf, nf + 2), np.float32)
        self.L.orc_tile_density(self.h, rank, tile[0], tile[1], tile[2], mass_p, rho)
        return rho

    def projection(self, mass_p):
        n = self.p.nf_physical_node_dim * self.p.nodes_dim
        maps = [np.empty((n, n), np.float32) for _ in range(3)]
        tot = C.c_double()
        self.L.orc_projection(self.h, mass_p, maps[0], maps[1], maps[2], C.byref(tot))
        return maps[0], maps[1], maps[2], tot.value

    def tile_force(self, rho):
        pt = self.p.nf_physical_tile_dim
        f = np.empty((pt + 3, pt + 3, pt + 3, 3), np.float32)
        m = C.c_float()
        self.L.orc_tile_force(self.h, np.ascontiguousarray(rho, np.float32), f, C.byref(m))
        return f, m.value

    def tile_velocity(self, rank, tile, f, a_mid, dt, mass_p):
        """gather + kick + intra-cell PP of one tile on the force box f [k][j][i][3]; -> (max |F|^2, pp_force_max)"""
        out = np.zeros(2, np.float32)
        self.L.orc_tile_velocity(self.h, rank, tile[0], tile[1], tile[2], np.ascontiguousarray(f, np.float32), a_mid, dt, mass_p, out)
        return float(out[0]), float(out[1])

    def coarse_density(self, mass_p):
        self.L.orc_coarse_density(self.h, mass_p)

    def rho_c(self, rank=0):
        n = self.p.nc_node_dim
        return np.ctypeslib.as_array(self.L.orc_rho_c(self.h, rank), shape=(n, n, n)).copy()

    def rho_c_view(self, rank=0):
        """writable view of the rank's coarse density (known-answer tests set it directly)"""
        n = self.p.nc_node_dim
        return np.ctypeslib.as_array(self.L.orc_rho_c(self.h, rank), shape=(n, n, n))

    def coarse_power(self, mass_p, box):
        """coarse_power.f90 on the current coarse density: (nc_dim, 2) rows (k, Delta^2(k)) of <z>ps.dat"""
        ps = np.zeros((self.p.nc_dim, 2), np.float32)
        self.L.orc_coarse_power(self.h, mass_p, box, ps)
        return ps

    def force_c(self, rank=0):
        n = self.p.nc_node_dim + 2
        return np.ctypeslib.as_array(self.L.orc_force_c(self.h, rank), shape=(n, n, n, 3)).copy()

    def coarse_force(self):
        self.L.orc_coarse_force(self.h)

    def distribute_force(self, fg):
        """fg: global coarse force [k][j][i][3] -> every rank's force_c incl. halo."""
        self.L.orc_distribute_force(self.h, np.ascontiguousarray(fg, np.float32))

    def coarse_max_dt_and_velocity(self, a_mid, dt):
        self.L.orc_coarse_max_dt_and_velocity(self.h, a_mid, dt)


class Oracle:
    """_RawOracle with the whole-step results memoised on disk (P3M_ORACLE_CACHE, default /tmp/p3m_oracle_cache; "0" turns it off).

    The GPU suite runs the same whole-step comparisons several times -- in the parent and again in child processes that select a
    run-time switch of the HIP library -- and the CPU oracle's particle_mesh (O(n^2) pair sums on one core) was most of the suite's wall
    time.  The chain  set_kernel_tables, set_particles..., particle_mesh..., get_particles  is keyed by a hash of the oracle's source and
    of every input; a hit returns the stored step_out / particles without running the C code.  Anything else that is called (phase
    calls, probes) first replays the logged calls on a real oracle, so the object always behaves like _RawOracle."""

    _src_hash = None

    def __init__(self, params: Params):
        import hashlib

        self.p = params
        self._raw, self._log, self._done, self._nocache = None, [], 0, os.environ.get("P3M_ORACLE_CACHE", "") == "0"
        self._dir = os.environ.get("P3M_ORACLE_CACHE") or "/tmp/p3m_oracle_cache"
        if Oracle._src_hash is None:
            # the key covers what decides the stored bytes: the BUILT oracle library (source, header, flags and compiler all end up in
            # it) and the layout of the step record the bytes are read back into (the source alone was not enough)
            hh = hashlib.sha1(open(build(), "rb").read())
            hh.update(str(C.sizeof(P3MStepOut)).encode())
            hh.update(open(os.path.join(ROOT, "include", "p3m_hip.h"), "rb").read())
            Oracle._src_hash = hh.hexdigest()
        self._h = hashlib.sha1((Oracle._src_hash + repr(bytes(params.to_c()))).encode())

    # -- plumbing
    def _feed(self, *parts):
        for a in parts:
            if a is None:
                self._h.update(b"<none>")
            elif isinstance(a, np.ndarray):
                self._h.update(str((a.dtype, a.shape)).encode()); self._h.update(np.ascontiguousarray(a).tobytes())
            else:
                self._h.update(repr(a).encode())

    def _real(self):
        if self._raw is None:
            self._raw = _RawOracle(self.p)
        while self._done < len(self._log):
            name, args = self._log[self._done]
            getattr(self._raw, name)(*args)
            self._done += 1
        return self._raw

    def _path(self, tag):
        return os.path.join(self._dir, self._h.hexdigest() + "_" + tag + ".npz")

    def _store(self, tag, **arrays):
        if self._nocache:
            return
        try:
            os.makedirs(self._dir, exist_ok=True)
            tmp = self._path(tag) + ".%d.tmp.npz" % os.getpid()
            np.savez(tmp, **arrays)
            os.replace(tmp, self._path(tag))
        except OSError:
            pass

    def _load(self, tag):
        if self._nocache or not os.path.exists(self._path(tag)):
            return None
        try:
            return np.load(self._path(tag))
        except Exception:
            return None

    def __getattr__(self, name):   # everything that is not memoised: a real oracle in the logged state; no caching afterwards
        if name.startswith("_"):
            raise AttributeError(name)
        raw = self._real()
        self._nocache = True
        return getattr(raw, name)

    def close(self):
        if self._raw is not None:
            self._raw.close()
            self._raw = None

    # -- the memoised chain
    def set_kernel_tables(self, fine, coarse):
        fine, coarse = np.ascontiguousarray(fine, np.float32), np.ascontiguousarray(coarse, np.float32)
        self._feed("kern", fine, coarse)
        self._log.append(("set_kernel_tables", (fine, coarse)))

    def set_particles(self, rank, xv, pid=None):
        xv = np.array(xv, np.float32, order="C")
        pid = None if pid is None else np.array(pid, np.int64, order="C")
        self._feed("set", rank, xv, pid)
        self._log.append(("set_particles", (rank, xv, pid)))

    def particle_mesh(self, a_mid, dt, dt_old, mass_p, offset=None, move_back=None):
        of = None if offset is None else np.array(offset, np.float32)
        mb = None if move_back is None else np.array(move_back, np.float32)
        self._feed("step", float(np.float32(a_mid)), float(np.float32(dt)), float(np.float32(dt_old)), float(np.float32(mass_p)), of, mb)
        hit = self._load("step")
        if hit is not None:
            self._log.append(("particle_mesh", (a_mid, dt, dt_old, mass_p, of, mb)))
            return P3MStepOut.from_buffer_copy(hit["out"].tobytes())
        raw = self._real()
        o = raw.particle_mesh(a_mid, dt, dt_old, mass_p, of, mb)
        self._log.append(("particle_mesh", (a_mid, dt, dt_old, mass_p, of, mb))); self._done = len(self._log)
        self._store("step", out=np.frombuffer(bytes(o), np.uint8))
        return o

    def get_particles(self, rank=0):
        hit = self._load("get%d" % rank)
        if hit is not None:
            return hit["xv"].copy(), hit["pid"].copy()
        xv, pid = self._real().get_particles(rank)
        self._store("get%d" % rank, xv=xv, pid=pid)
        return xv, pid


def fft3d(a, n, direction):
    a = np.ascontiguousarray(a, np.float32)
    lib().orc_fft3d(a, n, direction)
    return a


# ---- host time loop (timestep.f90): oracle side, same structs as the C ABI -------------------------------
def oracle_time_api():
    from cubep3m_amd.timestep import P3MTimeParams, P3MTimeState

    L = lib()
    L.orc_expansion.argtypes = [C.POINTER(P3MTimeParams), C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_expansion.restype = None
    L.orc_timestep.argtypes = [C.POINTER(P3MTimeParams), C.c_uint, C.POINTER(P3MTimeState)] + [C.c_float] * 4
    L.orc_timestep.restype = None
    return L
