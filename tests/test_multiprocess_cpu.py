"""world_size-2 `gloo` test on CPU of the host-side multi-rank logic: which process drives which logical
rank, and how a global particle set is cut into the reference's cubic sub-volumes (rank = c1*nd^2 + c2*nd +
c3, x <-> c3; mpi_initialization.f90:42-76).  The expected per-rank inputs are the ones the REFERENCE 8-rank
run of tests/golden/ref_stages_8rank.npz was fed with.  (The device-side exchanges need a GPU:
tests/test_gpu_group.py.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import cfg1
    from cubep3m_amd.group import local_ranks_of, owner_of_rank, split_global

    p = cfg1(nodes_dim=2)
    d = np.load(os.path.join(HERE, "golden", "ref_stages_8rank.npz"))
    Nn = p.nf_physical_node_dim
    # rebuild the global set the fixture was cut from
    xs, ps = [], []
    for r in range(8):
        c1, c2, c3 = r // 4, (r // 2) % 2, r % 2
        x = d["r%d_xv_in" % r].copy()
        x[:, :3] += np.array([c3, c2, c1], np.float32) * Nn
        xs.append(x)
        ps.append(d["r%d_pid_in" % r])
    xv, pid = np.concatenate(xs), np.concatenate(ps)
    perm = np.random.default_rng(3).permutation(len(xv))   # the global order must not matter
    xv, pid = xv[perm], pid[perm]
    mine = local_ranks_of(rank, p.nodes, world)
    assert all(owner_of_rank(r, p.nodes, world) == rank for r in mine)
    parts = split_global(p, xv, pid, mine)
    ok = True
    for r in mine:
        loc, lp = parts[r]
        o = np.argsort(lp)
        ref_o = np.argsort(d["r%d_pid_in" % r])
        ok &= np.array_equal(lp[o], d["r%d_pid_in" % r][ref_o])
        ok &= np.array_equal(loc[o], d["r%d_xv_in" % r][ref_o])     # bit-identical local coordinates
        ok &= bool(np.all((loc[:, :3] >= 0) & (loc[:, :3] < Nn)))
    # every particle is owned exactly once across the processes
    cnt = torch.tensor([sum(len(parts[r][1]) for r in mine)], dtype=torch.int64)
    dist.all_reduce(cnt)
    ids = [None] * world
    dist.all_gather_object(ids, sorted(mine))
    flat = sorted(sum(ids, []))
    okt = torch.tensor([1 if ok else 0])
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put((int(cnt.item()), len(xv), flat, int(okt.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_rank_ownership_and_domain_split_gloo(world):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    total, n, flat, ok = q.get(timeout=180)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert total == n and flat == list(range(8)) and ok == 1
