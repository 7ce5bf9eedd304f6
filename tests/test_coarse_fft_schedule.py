"""The exchange schedule of the distributed coarse transform (p3m_hip_coarse_fft_schedule: cube <-> x-lines, x <-> y, y <-> z)
for the slab decomposition (fftw3ds.f90:24-99) and the pencil decomposition (p3dfft_coarse.f90:69-183), checked on CPU: a
numpy model moves the blocks exactly where the schedule says -- inside one process, and between two `gloo` processes that own
half of the logical ranks each -- transforms one axis per stage, and must arrive at np.fft.rfftn of the global mesh with
every rank holding the wavenumbers the device code assumes (ky slab yz_index*s.., kx chunks xy_index*ncl..)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FLAG_PENCIL = 1 << 5


def schedule(L, nd, flags, rank, which):
    pe, ix = C.c_int32(), C.c_int32()
    n = L.p3m_hip_coarse_fft_schedule(nd, flags, rank, which, 0, C.byref(pe), C.byref(ix))
    assert n >= 0
    out = []
    for j in range(n):
        assert L.p3m_hip_coarse_fft_schedule(nd, flags, rank, which, j, C.byref(pe), C.byref(ix)) == n
        out.append((pe.value, ix.value))
    return out


class Model:
    """All logical ranks of `mine` live here; route(msgs) delivers {(dst, index): block} produced by every process."""

    def __init__(self, L, nd, ncn, pencil, mine, route):
        self.L, self.nd, self.ncn, self.pencil, self.mine, self.route = L, nd, ncn, pencil, mine, route
        self.flags = FLAG_PENCIL if pencil else 0
        self.nc = ncn * nd
        self.s = ncn // nd if pencil else self.nc // nd ** 3
        h = self.nc // 2 + 1
        unit = 16 * nd if pencil else 16
        self.px = (h + unit - 1) // unit * unit
        self.pxl = self.px // nd if pencil else self.px

    def exchange(self, which, blocks_of):
        """blocks_of(r) -> list of blocks, block j goes to peer j; returns {r: [blocks ordered by index]}"""
        msgs = {}
        for r in self.mine:
            sch = schedule(self.L, self.nd, self.flags, r, which)
            bl = blocks_of(r)
            assert len(bl) == len(sch)
            for (peer, index), b in zip(sch, bl):
                assert (peer, index) not in msgs
                msgs[(peer, index)] = b
        got = self.route(msgs)
        n = len(schedule(self.L, self.nd, self.flags, self.mine[0], which))
        return {r: [got[(r, i)] for i in range(n)] for r in self.mine}

    def forward(self, rho):
        nd, ncn, nc, s = self.nd, self.ncn, self.nc, self.s
        cube = {}
        for r in self.mine:
            c1, c2, c3 = r // (nd * nd), (r // nd) % nd, r % nd
            cube[r] = rho[c1 * ncn:(c1 + 1) * ncn, c2 * ncn:(c2 + 1) * ncn, c3 * ncn:(c3 + 1) * ncn]
        # 0: z-slice q of the cube -> peer q; arrivals are blocks (j*nd + i) of the x-lines (pencils: i only)
        arr = self.exchange(0, lambda r: [cube[r][q * s:(q + 1) * s] for q in range(len(schedule(self.L, nd, self.flags, r, 0)))])
        lines = {}
        for r in self.mine:
            if self.pencil:
                lines[r] = np.concatenate(arr[r], axis=2)                                   # [s][ncn][nc]
            else:
                lines[r] = np.concatenate([np.concatenate(arr[r][j * nd:(j + 1) * nd], axis=2) for j in range(nd)], axis=1)   # [s][nc][nc]
            hat = np.fft.rfft(lines[r].astype(np.float64), axis=2)
            pad = np.zeros(hat.shape[:2] + (self.px,), np.complex128)
            pad[:, :, :hat.shape[2]] = hat
            lines[r] = pad
        if self.pencil:                                                                      # 1: kx chunk range j -> peer j, arrivals stack along y
            arr = self.exchange(1, lambda r: [lines[r][:, :, j * self.pxl:(j + 1) * self.pxl] for j in range(nd)])
            lines = {r: np.concatenate(arr[r], axis=1) for r in self.mine}                   # [s][nc][pxl]
        for r in self.mine:
            lines[r] = np.fft.fft(lines[r], axis=1)
        # 2: ky block j -> peer j, arrivals stack along z
        nyz = len(schedule(self.L, nd, self.flags, self.mine[0], 2))
        assert nyz * s == nc
        arr = self.exchange(2, lambda r: [lines[r][:, j * s:(j + 1) * s, :] for j in range(nyz)])
        out = {}
        for r in self.mine:
            out[r] = np.fft.fft(np.concatenate(arr[r], axis=0), axis=0)                      # [nc kz][s ky][pxl kx]
        return out

    def check(self, out, rho):
        want = np.fft.rfftn(rho.astype(np.float64))                                          # [kz][ky][kx]
        h = self.nc // 2 + 1
        covered = np.zeros((self.nc, h), bool)
        for r in self.mine:
            yz = schedule(self.L, self.nd, self.flags, r, 2)[0][1]
            kx0 = schedule(self.L, self.nd, self.flags, r, 1)[0][1] * self.pxl if self.pencil else 0
            nkx = max(0, min(h - kx0, self.pxl))
            got = out[r][:, :, :nkx]
            ref = want[:, yz * self.s:(yz + 1) * self.s, kx0:kx0 + nkx]
            if nkx:   # a rank whose chunks are all pad columns holds zeros only
                assert np.abs(got - ref).max() <= 1e-9 * np.abs(want).max(), r
            assert np.all(out[r][:, :, nkx:] == 0)                                           # pad columns
            covered[yz * self.s:(yz + 1) * self.s, kx0:kx0 + nkx] = True
        return covered


@pytest.mark.parametrize("nd,ncn,pencil", [(2, 16, False), (2, 16, True), (3, 12, True), (2, 6, True), (1, 8, False), (3, 9, False)])
def test_schedule_composes_into_the_global_transform(nd, ncn, pencil):
    sys.path.insert(0, ROOT)
    from cubep3m_amd import lib

    L = lib.load()
    rho = np.random.default_rng(7).random((ncn * nd,) * 3, dtype=np.float32)
    m = Model(L, nd, ncn, pencil, list(range(nd ** 3)), lambda msgs: msgs)
    covered = m.check(m.forward(rho), rho)
    assert covered.all()   # every (ky, kx) is held by exactly the rank the index maps name


@pytest.mark.parametrize("nd", [2, 3, 4])
def test_pencil_pack_partners_are_the_references(nd):
    """pen_neighbor_to(j) / pen_neighbor_fm(j) of mpi_initialization_p3dfft.f90:43-51, restated from the text:
    slab_coord(3) = rank / nd^2, slab_coord(2) = (rank mod nd^2) / nd, slab_coord(1) = rank mod nd,
    to(j) = nd^2*sc(3) + sc(2) + j*nd, fm(j) = nd^2*sc(3) + j + nd*sc(1); slice j goes to to(j), block j comes from fm(j)."""
    sys.path.insert(0, ROOT)
    from cubep3m_amd import lib

    L = lib.load()
    for r in range(nd ** 3):
        sc3, sc2, sc1 = r // (nd * nd), (r % (nd * nd)) // nd, r % nd
        sch = schedule(L, nd, FLAG_PENCIL, r, 0)
        assert [pe for pe, _ in sch] == [nd * nd * sc3 + sc2 + j * nd for j in range(nd)]
        for j in range(nd):
            fm = nd * nd * sc3 + j + nd * sc1
            back = schedule(L, nd, FLAG_PENCIL, fm, 0)
            assert r in [pe for pe, _ in back] and back[0][1] == j   # fm(j) sends to r, and lands as x-block j


def test_schedule_rejects_what_does_not_exist():
    sys.path.insert(0, ROOT)
    from cubep3m_amd import lib

    L = lib.load()
    pe, ix = C.c_int32(), C.c_int32()
    assert L.p3m_hip_coarse_fft_schedule(2, 0, 0, 1, 0, C.byref(pe), C.byref(ix)) == 0       # slabs have no x<->y exchange
    assert L.p3m_hip_coarse_fft_schedule(2, 0, 8, 0, 0, C.byref(pe), C.byref(ix)) < 0        # rank out of range
    assert L.p3m_hip_coarse_fft_schedule(2, FLAG_PENCIL, 0, 2, 4, C.byref(pe), C.byref(ix)) < 0   # 4 peers only
    assert L.p3m_hip_coarse_fft_schedule(2, 0, 0, 3, 0, C.byref(pe), C.byref(ix)) < 0


def _worker(rank, world, port, pencil, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cubep3m_amd import lib
    from cubep3m_amd.group import local_ranks_of, owner_of_rank

    L = lib.load()
    nd, ncn = 2, 16
    mine = local_ranks_of(rank, nd ** 3, world)

    def route(msgs):   # every process hands over what it sends; each keeps what is addressed to its own ranks
        allm = [None] * world
        dist.all_gather_object(allm, {k: v for k, v in msgs.items() if owner_of_rank(k[0], nd ** 3, world) != rank})
        got = {k: v for k, v in msgs.items() if owner_of_rank(k[0], nd ** 3, world) == rank}
        for m in allm:
            for k, v in m.items():
                if owner_of_rank(k[0], nd ** 3, world) == rank:
                    assert k not in got
                    got[k] = v
        return got

    rho = np.random.default_rng(11).random((ncn * nd,) * 3, dtype=np.float32)
    m = Model(L, nd, ncn, pencil, mine, route)
    covered = m.check(m.forward(rho), rho)
    allc = [None] * world
    dist.all_gather_object(allc, covered)
    ok = bool(np.logical_or.reduce(allc).all()) and sum(int(c.sum()) for c in allc) == covered.size   # disjoint and complete
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("pencil", [False, True])
def test_schedule_between_two_gloo_processes(pencil):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pencil, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
