"""Multi-rank host: the reference's nodes_dim^3 MPI ranks as a group of logical ranks spread over the
GPUs of one node (include/p3m_hip.h, p3m_hip_group_*).  One process drives one GPU.

    g = ParticleMeshGroup(Params(nodes_dim=2, ...), proc=rank, nprocs=world, unique_id=uid)
    g.scatter_global(xv_global, pid)      # rank r gets the particles of its cube, in local coordinates
    out = g.particle_mesh(a_mid, dt, dt_old, mass_p)

Rank numbering and coordinates follow mpi_initialization.f90:42-76: rank = c1*nd^2 + c2*nd + c3 with
x <-> c3, y <-> c2, z <-> c1.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .kernels import default_tables
from .params import P3MStepOut, Params


def rccl_unique_id() -> bytes:
    """128-byte ncclUniqueId; call on process 0 and broadcast (torch.distributed / MPI_Bcast)."""
    buf = (C.c_ubyte * 128)()
    _lib.check(_lib.load().p3m_hip_rccl_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


# ---- host transport (struct p3m_transport of include/p3m_hip.h) -----------------------------------
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64))
ALLREDUCE_F32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int32)
ALLREDUCE_F64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


class P3MTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", EXCHANGE_FN), ("allreduce_max_f32", ALLREDUCE_F32_FN),
                ("allreduce_sum_f64", ALLREDUCE_F64_FN)]


def torch_transport(dist, group=None) -> P3MTransport:
    """The three callbacks over a CPU-tensor torch.distributed backend (gloo): what an MPI host implements with
    MPI_Irecv/MPI_Isend/MPI_Waitall and MPI_Allreduce.  Used where RCCL cannot connect the processes
    (e.g. two processes sharing one GPU in the tests)."""
    import traceback

    import torch

    def _view(ptr, nbytes):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)))

    def exchange(user, n, peer, sbuf, sbytes, rbuf, rbytes):
        try:
            works = []
            for i in range(n):
                if rbytes[i] > 0:
                    works.append(dist.irecv(_view(rbuf[i], rbytes[i]), src=int(peer[i]), group=group))
            for i in range(n):
                if sbytes[i] > 0:
                    works.append(dist.isend(_view(sbuf[i], sbytes[i]), dst=int(peer[i]), group=group))
            for w in works:
                w.wait()
            return 0
        except Exception:  # an exception must not unwind through the C frames
            traceback.print_exc()
            return 1

    def _allreduce(ctype, op):
        def fn(user, v, n):
            try:
                t = torch.from_numpy(np.ctypeslib.as_array(v, shape=(int(n),)))
                dist.all_reduce(t, op=op, group=group)
                return 0
            except Exception:
                traceback.print_exc()
                return 1
        return fn

    return P3MTransport(None, EXCHANGE_FN(exchange), ALLREDUCE_F32_FN(_allreduce(C.c_float, dist.ReduceOp.MAX)),
                        ALLREDUCE_F64_FN(_allreduce(C.c_double, dist.ReduceOp.SUM)))


def rank_coords(rank, nd):
    return rank // (nd * nd), (rank // nd) % nd, rank % nd  # c1 (z), c2 (y), c3 (x)


def owner_of_rank(rank, nodes, nprocs):
    """Process that drives logical rank `rank`: contiguous blocks, so whole z-layers share a GPU where possible
    (same rule as p3m_hip_group_create)."""
    return rank // (nodes // nprocs)


def local_ranks_of(proc, nodes, nprocs):
    per = nodes // nprocs
    return list(range(proc * per, (proc + 1) * per))


def split_global(params: Params, xv, pid, ranks):
    """Particles of the global periodic box -> {logical rank: (xv in the rank's local coordinates, pid)}."""
    nd, Nn = params.nodes_dim, params.nf_physical_node_dim
    out = {}
    cell = np.floor(xv[:, :3] / np.float32(Nn)).astype(np.int64)
    owner = cell[:, 2] * nd * nd + cell[:, 1] * nd + cell[:, 0]
    for r in ranks:
        m = owner == r
        c1, c2, c3 = rank_coords(r, nd)
        loc = xv[m].copy()
        loc[:, :3] -= np.array([c3, c2, c1], np.float32) * np.float32(Nn)
        out[r] = (loc, pid[m])
    return out


class ParticleMeshGroup:
    def __init__(self, params: Params, proc=0, nprocs=1, fine_table=None, coarse_table=None, unique_id=None,
                 force_rccl=False, set_kernels=True, transport: "P3MTransport | None" = None):
        self.params = params
        self.L = _lib.load()
        self._cp = params.to_c()
        h = C.c_void_p()
        _lib.check(self.L.p3m_hip_group_create(C.byref(self._cp), proc, nprocs, C.byref(h)))
        self.h = h
        self.proc, self.nprocs = proc, nprocs
        if unique_id is not None:
            buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id)[:128].ljust(128, b"\0"))
            _lib.check(self.L.p3m_hip_group_comm_init_rccl(self.h, C.cast(buf, C.c_void_p), 1 if force_rccl else 0))
        if transport is not None:
            self._transport = transport   # keeps the ctypes callbacks alive as long as the group
            _lib.check(self.L.p3m_hip_group_set_transport(self.h, C.byref(transport)))
        self.nlocal = self.L.p3m_hip_group_nlocal(self.h)
        self.local_ranks = [self.L.p3m_hip_group_local_rank(self.h, i) for i in range(self.nlocal)]
        if set_kernels:
            self.set_kernel_tables(fine_table, coarse_table)

    def set_kernel_tables(self, fine_table=None, coarse_table=None):
        """fine_kernel / coarse_kernel from the two ascii tables (collective over the processes of the group)."""
        if fine_table is None or coarse_table is None:
            fine_table, coarse_table = default_tables()
        _lib.check(self.L.p3m_hip_group_set_kernel_tables(self.h, np.ascontiguousarray(fine_table, np.float32),
                                                          np.ascontiguousarray(coarse_table, np.float32)))

    def set_kernels_raw(self, kern_f, kern_c_slabs):
        """kern_f (nf, nf, nf/2+1, 3) and, per LOCAL rank, its z-slab of kern_c (nc_slab, nc, nc/2+1, 3) -- the reference's arrays
        (component fastest); collective over the processes of the group."""
        kf = np.ascontiguousarray(kern_f, np.float32)
        slabs = [np.ascontiguousarray(k, np.float32) for k in kern_c_slabs]
        assert len(slabs) == self.nlocal
        arr = (C.c_void_p * self.nlocal)(*[k.ctypes.data for k in slabs])
        _lib.check(self.L.p3m_hip_group_set_kernels_raw(self.h, kf, arr))

    def comm_info(self):
        """{"comm_count", "comm_rank"} of the RCCL communicator (-1: none), the HIP device ordinal and its UUID."""
        cnt, rk, dev = C.c_int32(), C.c_int32(), C.c_int32()
        uuid = C.create_string_buffer(33)
        _lib.check(self.L.p3m_hip_group_comm_info(self.h, C.byref(cnt), C.byref(rk), C.byref(dev), uuid))
        return {"comm_count": cnt.value, "comm_rank": rk.value, "device": dev.value, "uuid": uuid.value.decode()}

    def coarse_power(self, mass_p, box):
        """coarse_power.f90 on the coarse density of the last particle_mesh step: (nc_dim, 2) rows (k, Delta^2(k))."""
        ps = np.zeros((self.params.nc_dim, 2), np.float32)
        _lib.check(self.L.p3m_hip_group_coarse_power(self.h, mass_p, box, ps))
        return ps

    def close(self):
        if getattr(self, "h", None):
            self.L.p3m_hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- particles -------------------------------------------------------------------------
    def upload_particles(self, i, xv, pid=None):
        xv = np.ascontiguousarray(xv, np.float32).reshape(-1, 6)
        pp = None
        if pid is not None:
            pid = np.ascontiguousarray(pid, np.int64)
            pp = pid.ctypes.data_as(C.c_void_p)
        _lib.check(self.L.p3m_hip_group_upload_particles(self.h, i, xv.ctypes.data_as(C.c_void_p), pp, len(xv)))

    def download_particles(self, i):
        n = C.c_int32()
        _lib.check(self.L.p3m_hip_group_download_particles(self.h, i, None, None, C.byref(n)))
        xv = np.empty((n.value, 6), np.float32)
        pid = np.empty(n.value, np.int64)
        _lib.check(self.L.p3m_hip_group_download_particles(self.h, i, xv.ctypes.data_as(C.c_void_p), pid.ctypes.data_as(C.c_void_p), C.byref(n)))
        return xv, pid

    def split_global(self, xv, pid):
        """Particles of the global box -> {logical rank: (xv_local, pid)} for the ranks this process owns."""
        return split_global(self.params, xv, pid, self.local_ranks)

    def scatter_global(self, xv, pid):
        parts = self.split_global(xv, pid)
        for i, r in enumerate(self.local_ranks):
            self.upload_particles(i, *parts[r])
        return parts

    # -- subroutine particle_mesh on every rank ----------------------------------------------
    def particle_mesh(self, a_mid, dt, dt_old, mass_p, offset=None, move_back=None) -> P3MStepOut:
        o = P3MStepOut()
        po = None if offset is None else np.ascontiguousarray(offset, np.float32)
        pm = None if move_back is None else np.ascontiguousarray(move_back, np.float32)
        _lib.check(self.L.p3m_hip_group_particle_mesh(self.h, a_mid, dt, dt_old, mass_p,
                                                      None if po is None else po.ctypes.data_as(C.c_void_p),
                                                      None if pm is None else pm.ctypes.data_as(C.c_void_p), C.byref(o)))
        return o

    PHASES = ("update_position", "link_list", "particle_pass", "fine_mass", "fine_fft", "fine_kick", "pp_intra", "pp_ext",
              "coarse_mass", "coarse_force", "coarse_velocity", "delete_particles")

    def phase_timing(self, on=True):
        """per-phase GPU times of the following steps (timers.f90:68-77, -DMPI_TIME)"""
        _lib.check(self.L.p3m_hip_group_phase_timing(self.h, 1 if on else 0))

    def last_phase_ms(self):
        ms = np.zeros(12, np.float32)
        _lib.check(self.L.p3m_hip_group_last_phase_ms(self.h, ms))
        return dict(zip(self.PHASES, (float(v) for v in ms)))

    def update_position(self, dt, dt_old, offset=None):
        po = None if offset is None else np.ascontiguousarray(offset, np.float32)
        _lib.check(self.L.p3m_hip_group_update_position(self.h, dt, dt_old, None if po is None else po.ctypes.data_as(C.c_void_p)))

    def rank_context(self, i):
        """The i-th local rank's context as a ParticleMesh (probes, kernel timers)."""
        from .particle_mesh import ParticleMesh

        return ParticleMesh.from_handle(self.L.p3m_hip_group_ctx(self.h, i), self.params)

    def projection(self, mass_p):
        """projection.f90 at a projection step (ghost pass, sort, CIC projection, ghost removal): the maps summed over this
        process's logical ranks and the projected mass summed over all ranks; several processes add their maps up."""
        n = self.params.nf_physical_node_dim * self.params.nodes_dim
        maps = [np.empty((n, n), np.float32) for _ in range(3)]
        tot = C.c_double()
        _lib.check(self.L.p3m_hip_group_projection(self.h, mass_p, *(m.ctypes.data_as(C.c_void_p) for m in maps), C.byref(tot)))
        return maps[0], maps[1], maps[2], tot.value

    # -- the distributed coarse transform on its own (any multi-rank group; Params(coarse_only=True) holds nothing else) -----
    def set_coarse_density(self, i, rho_c):
        n = self.params.nc_node_dim
        rho_c = np.ascontiguousarray(rho_c, np.float32)
        assert rho_c.shape == (n, n, n)
        _lib.check(self.L.p3m_hip_group_set_coarse_density(self.h, i, rho_c))

    def coarse_transform(self, what, reps=0):
        """what = "forward": the distributed r2c transform of every rank's rho_c; "force": coarse_force.f90 (forward, multiply,
        three inverse transforms, force halo).  Returns the average ms per run over `reps` timed runs (None for reps = 0)."""
        ms = C.c_float()
        _lib.check(self.L.p3m_hip_group_coarse_transform(self.h, {"forward": 0, "force": 1}[what], reps, C.byref(ms)))
        return ms.value if reps > 0 else None

    def coarse_hat(self, i):
        """rho-hat of local rank i as complex64 [local ky][kz][kx] (kx up to the padded row pitch; slabs: ky = rank*nc_slab + local ky)."""
        p = self.params
        px = ((p.nc_dim // 2 + 1) + 15) // 16 * 16
        if p.pencil:
            px = (px + 16 * p.nodes_dim - 1) // (16 * p.nodes_dim) * (16 * p.nodes_dim)
        s = p.nc_node_dim // p.nodes_dim if p.pencil else p.nc_slab
        ncl = px // 16 // (p.nodes_dim if p.pencil else 1)
        raw = np.empty((s, ncl, p.nc_dim, 16), np.complex64)
        _lib.check(self.L.p3m_hip_group_get_coarse_hat(self.h, i, raw.ctypes.data_as(C.c_void_p), raw.size * 2))
        return raw.transpose(0, 2, 1, 3).reshape(s, p.nc_dim, ncl * 16)

    def coarse_force(self, i):
        n = self.params.nc_node_dim + 2
        f = np.empty((n, n, n, 3), np.float32)
        _lib.check(self.L.p3m_hip_group_get_coarse_force(self.h, i, f.ctypes.data_as(C.c_void_p)))
        return f

    @property
    def coarse_exchange_bytes(self):
        return int(self.L.p3m_hip_group_coarse_exchange_bytes(self.h))

    def coarse(self, mass_p, i, want_force=True):
        p = self.params
        rho = np.empty((p.nc_node_dim,) * 3, np.float32)
        f = np.empty((p.nc_node_dim + 2,) * 3 + (3,), np.float32) if want_force else None
        _lib.check(self.L.p3m_hip_group_probe_coarse(self.h, mass_p, i, rho.ctypes.data_as(C.c_void_p),
                                                     f.ctypes.data_as(C.c_void_p) if want_force else None))
        return rho, f
