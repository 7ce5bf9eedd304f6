"""GPU parity of the multi-rank path: 2x2x2 logical ranks on ONE GPU (device-to-device exchanges, and
again with every exchange forced through RCCL send/recv to self) against the oracle's 8 simulated ranks."""
import numpy as np
import pytest

import oracle_lib as ol
from common import observed, COARSE_TABLE, FINE_TABLE, by_pid, cfg1, clustered_particles, rel_rms, uniform_particles

# multi-step tests: errors of step 1 feed step 2's drift.  Rounds 1-3 held them to 2e-5 / 3e-5 without recording what was observed;
# common.observed() now records it (gpurun_out/observed_errors.txt, tabulated in DESIGN section 4): 2e-7 ... 2e-6 over two steps,
# <= 4.7e-6 over five -- so they are held to the single-step bar, 1e-5, like everything else
BAR_KICK_2STEP, BAR_KICK_5STEP, BAR_DT_5STEP = 1e-5, 1e-5, 1e-5

pytestmark = pytest.mark.gpu


def global_ic(kind, n, box, seed):
    xv = uniform_particles(n, box, seed=seed) if kind == "uniform" else clustered_particles(n, box, seed=seed, frac=0.3, nblobs=40, sigma=0.7, vel_sigma=0.5)
    pid = np.arange(1, n + 1, dtype=np.int64) * 3 + 1
    return xv, pid


def run_both(p, xv, pid, scal, force_rccl=False, steps=1):
    from cubep3m_amd.group import ParticleMeshGroup, rccl_unique_id

    uid = rccl_unique_id() if force_rccl else None
    g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE, unique_id=uid, force_rccl=force_rccl)
    assert g.nlocal == p.nodes and g.local_ranks == list(range(p.nodes))
    parts = g.scatter_global(xv, pid)
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    for r in range(p.nodes):
        o.set_particles(r, *parts[r])
    a_mid, dt, dt_old, mass_p = scal
    for s in range(steps):
        og = g.particle_mesh(a_mid, dt, dt_old if s == 0 else dt, mass_p)
        oo = o.particle_mesh(a_mid, dt, dt_old if s == 0 else dt, mass_p)
    return g, o, og, oo


@pytest.mark.parametrize("kind,kw,force_rccl", [
    ("uniform", dict(ngp=True), False),
    ("clustered", dict(ngp=True, ppint=True, pp_ext=True), False),
    ("clustered", dict(ngp=True, ppint=True, pp_ext=True, lrckcorr=True), False),
    ("uniform", dict(ngp=False), True),
    ("clustered", dict(ngp=True, ppint=True, pp_ext=True), True),
    ("clustered", dict(ngp=True, ppint=True, pp_ext=True, lrckcorr=True, pencil=True), False),
    ("uniform", dict(ngp=False, pencil=True), True),
    ("clustered", dict(ngp=True, coarse_ngp=True), False),          # -DCOARSE_NGP across rank boundaries (force halo cells)
])
def test_eight_logical_ranks_match_oracle(kind, kw, force_rccl):
    p = cfg1(nodes_dim=2, **kw)
    box = float(p.nf_physical_dim)  # 128
    xv, pid = global_ic(kind, 60000, box, 99)
    # v != 0 and dt_old != 0: particles cross rank boundaries in the drift
    g, o, og, oo = run_both(p, xv, pid, (0.01, 0.3, 0.3, 8.0), force_rccl=force_rccl)
    assert og.np_total == oo.np_total == len(xv)
    assert og.np_ghost == oo.np_ghost and og.np_deleted == oo.np_deleted
    for name in ("dt_f_acc", "dt_c_acc") + (("dt_pp_acc", "dt_pp_ext_acc") if kw.get("pp_ext") else ()):
        assert getattr(og, name) == pytest.approx(getattr(oo, name), rel=1e-5), name
    assert og.sum_rho_f == pytest.approx(oo.sum_rho_f, rel=1e-6) and og.sum_rho_c == pytest.approx(oo.sum_rho_c, rel=1e-6)
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 1e-4
        vin = np.stack([v0[q] for q in pg.tolist()])
        dg, do = xg[:, 3:].astype(np.float64) - vin, xo[:, 3:].astype(np.float64) - vin
        num += ((dg - do) ** 2).sum()
        den += (do ** 2).sum()
    assert np.sqrt(num / den) <= 1e-5, np.sqrt(num / den)


@pytest.mark.parametrize("pencil", [False, True])
def test_distributed_coarse_mesh_vs_oracle(pencil):
    """Slab FFT (fftw3ds.f90) with the all-to-all transpose, cube<->slab redistribution and the force halo, rank by rank; and
    the same through the pencil decomposition (p3dfft_coarse.f90: x-pencils, x<->y and y<->z transposes)."""
    from cubep3m_amd.group import ParticleMeshGroup

    p = cfg1(nodes_dim=2, lrckcorr=True, pencil=pencil)
    xv, pid = global_ic("clustered", 40000, float(p.nf_physical_dim), 5)
    g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE)
    parts = g.scatter_global(xv, pid)
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    for r in range(8):
        o.set_particles(r, *parts[r])
    # dt = dt_old = 0: no drift; the step leaves the cell-sorted records (with ghosts) on the device for the probe
    g.particle_mesh(0.01, 0.0, 0.0, 8.0)
    o.link_list()
    assert o.particle_pass() == 0
    o.coarse_density(8.0)
    o.coarse_force()
    for i in range(8):
        rg, fg = g.coarse(8.0, i)
        ro, fo = o.rho_c(i), o.force_c(i)
        assert np.abs(rg - ro).max() <= 4e-6 * np.abs(ro).max(), i
        assert rel_rms(fg, fo) < 3e-6, i      # interior and the one-cell halo from the neighbours


def test_pencil_and_slab_decompositions_give_the_same_coarse_force():
    """Both decompositions run the same line transforms on the same lines, only in different places: the coarse force and
    the spectrum agree to rounding."""
    from cubep3m_amd.group import ParticleMeshGroup

    res = {}
    for pencil in (False, True):
        p = cfg1(nodes_dim=2, lrckcorr=True, pencil=pencil)
        xv, pid = global_ic("clustered", 40000, float(p.nf_physical_dim), 5)
        g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE)
        g.scatter_global(xv, pid)
        g.particle_mesh(0.01, 0.0, 0.0, 8.0)
        res[pencil] = ([g.coarse(8.0, i)[1] for i in range(8)], g.coarse_power(8.0, 100.0))
        g.close()
    for a, b in zip(res[False][0], res[True][0]):
        assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()
    assert np.allclose(res[False][1], res[True][1], rtol=1e-6, atol=0)


def test_pencils_where_slabs_cannot_decompose_the_mesh():
    """3x3x3 ranks on a 36^3 coarse mesh: 36 is not a multiple of 27, so mpi_initialization.f90:26 refuses the slab build; the
    pencil build (nc_pen = 12/3 = 4 planes, kx chunks padded to 3 x 16) runs, and matches the oracle's 27 ranks."""
    from cubep3m_amd.group import ParticleMeshGroup
    from cubep3m_amd.lib import P3MError
    from cubep3m_amd.params import Params

    kw = dict(nodes_dim=3, tiles_node_dim=1, nf_tile=96, cores=1, ngp=True, ppint=True, pp_ext=True)
    with pytest.raises((ValueError, P3MError)):
        ParticleMeshGroup(Params(**kw), 0, 1, FINE_TABLE, COARSE_TABLE)
    p = Params(pencil=True, **kw)
    assert p.nc_dim == 36 and p.nc_dim % p.nodes != 0
    xv, pid = global_ic("clustered", 50000, float(p.nf_physical_dim), 17)
    g, o, og, oo = run_both(p, xv, pid, (0.01, 0.3, 0.3, 8.0))
    assert og.np_total == oo.np_total == len(xv) and og.np_ghost == oo.np_ghost
    for name in ("dt_f_acc", "dt_c_acc", "dt_pp_acc", "dt_pp_ext_acc"):
        assert getattr(og, name) == pytest.approx(getattr(oo, name), rel=1e-5), name
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        vin = np.stack([v0[q] for q in pg.tolist()]) if len(pg) else np.zeros((0, 3), np.float32)
        dg, do = xg[:, 3:].astype(np.float64) - vin, xo[:, 3:].astype(np.float64) - vin
        num += ((dg - do) ** 2).sum()
        den += (do ** 2).sum()
    assert np.sqrt(num / den) <= 1e-5, np.sqrt(num / den)


# ------------------------------------------------------------------ two PROCESSES (4 logical ranks each)
def _two_proc_worker(rank, world, port, outdir, kw):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cubep3m_amd.group import ParticleMeshGroup, torch_transport

    p = cfg1(nodes_dim=2, **kw)
    xv, pid = global_ic("clustered", 60000, float(p.nf_physical_dim), 99)
    g = ParticleMeshGroup(p, rank, world, FINE_TABLE, COARSE_TABLE, transport=torch_transport(dist))
    assert g.local_ranks == list(range(rank * 8 // world, (rank + 1) * 8 // world))
    g.scatter_global(xv, pid)
    out = g.particle_mesh(0.01, 0.3, 0.3, 8.0)
    res = {"np_total": out.np_total, "np_ghost": out.np_ghost, "np_deleted": out.np_deleted, "dt_f_acc": out.dt_f_acc, "dt_c_acc": out.dt_c_acc,
           "dt_pp_acc": out.dt_pp_acc, "dt_pp_ext_acc": out.dt_pp_ext_acc, "sum_rho_f": out.sum_rho_f, "sum_rho_c": out.sum_rho_c}
    for i, r in enumerate(g.local_ranks):
        x, q = g.download_particles(i)
        res["xv%d" % r], res["pid%d" % r] = x, q
    np.savez(os.path.join(outdir, "proc%d.npz" % rank), **res)
    dist.barrier()
    g.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pencil", [(2, False), (8, False), (2, True), (8, True)])
def test_several_processes_host_transport_match_oracle(tmp_path, world, pencil):
    """The process-level split (8/world logical ranks per process, remote peers, announced counts, the all-to-all and
    halo message lists, the dt reductions) driven by `world` processes sharing this one GPU through the host-callback
    transport over gloo -- RCCL refuses two ranks on one device, and an MPI host would take this same route.
    world = 8 is the shape of an 8-GPU run: one logical rank per process, every exchange leaves the process."""
    import socket

    import torch.multiprocessing as mp

    kw = dict(ngp=True, ppint=True, pp_ext=True, pencil=pencil)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_two_proc_worker, args=(r, world, port, str(tmp_path), kw)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
        assert pr.exitcode == 0
    p = cfg1(nodes_dim=2, **kw)
    xv, pid = global_ic("clustered", 60000, float(p.nf_physical_dim), 99)
    from cubep3m_amd.group import split_global

    parts = split_global(p, xv, pid, range(8))
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    for r in range(8):
        o.set_particles(r, *parts[r])
    oo = o.particle_mesh(0.01, 0.3, 0.3, 8.0)
    res = [np.load(tmp_path / ("proc%d.npz" % r)) for r in range(world)]
    ghosts = 0
    for d in res:   # the reductions reached both processes
        assert int(d["np_total"]) == oo.np_total == len(xv)
        for name in ("dt_f_acc", "dt_c_acc", "dt_pp_acc", "dt_pp_ext_acc"):
            assert float(d[name]) == pytest.approx(getattr(oo, name), rel=1e-5), name
        assert float(d["sum_rho_f"]) == pytest.approx(oo.sum_rho_f, rel=1e-6) and float(d["sum_rho_c"]) == pytest.approx(oo.sum_rho_c, rel=1e-6)
        ghosts += int(d["np_ghost"])
    assert ghosts == oo.np_ghost
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for r in range(8):
        d = res[r // (8 // world)]
        xg, pg = by_pid(d["xv%d" % r], d["pid%d" % r])
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 1e-4
        vin = np.stack([v0[q] for q in pg.tolist()])
        dg, do = xg[:, 3:].astype(np.float64) - vin, xo[:, 3:].astype(np.float64) - vin
        num += ((dg - do) ** 2).sum()
        den += (do ** 2).sum()
    assert np.sqrt(num / den) <= 1e-5, np.sqrt(num / den)


def test_eight_logical_ranks_with_mesh_offsets_and_move_grid_back():
    """DISP_MESH offset in the drift and MOVE_GRID_BACK before the ghost removal, across rank boundaries."""
    from cubep3m_amd.group import ParticleMeshGroup

    p = cfg1(nodes_dim=2, ngp=True, ppint=True, pp_ext=True, move_grid_back=True)
    xv, pid = global_ic("clustered", 50000, float(p.nf_physical_dim), 21)
    g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE)
    parts = g.scatter_global(xv, pid)
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    for r in range(8):
        o.set_particles(r, *parts[r])
    rng = np.random.default_rng(5)
    for step in range(2):
        off = ((rng.random(3, dtype=np.float32) - np.float32(0.5)) * np.float32(16.0)).astype(np.float32)
        og = g.particle_mesh(0.02, 0.25, 0.25, 8.0, offset=off, move_back=off)
        oo = o.particle_mesh(0.02, 0.25, 0.25, 8.0, offset=off, move_back=off)
        assert og.np_total == oo.np_total == len(xv) and og.np_ghost == oo.np_ghost and og.np_deleted == oo.np_deleted, step
        for name in ("dt_f_acc", "dt_c_acc", "dt_pp_acc", "dt_pp_ext_acc"):
            assert getattr(og, name) == pytest.approx(getattr(oo, name), rel=1e-5), (step, name)
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 1e-4
        vin = np.stack([v0[q] for q in pg.tolist()])
        dg, do = xg[:, 3:].astype(np.float64) - vin, xo[:, 3:].astype(np.float64) - vin
        num += ((dg - do) ** 2).sum()
        den += (do ** 2).sum()
    assert observed("test_eight_logical_ranks_with_mesh_offsets_and_move_grid_back: kick, relative rms over both steps", np.sqrt(num / den), BAR_KICK_2STEP) <= BAR_KICK_2STEP


def test_eight_logical_ranks_one_tile_each_like_the_bench_workload():
    """The shape of the default bench workload (nodes_dim = 2, ONE tile per rank) at test size."""
    p = cfg1(nodes_dim=2, tiles_node_dim=1, nf_tile=112, ngp=True)
    xv, pid = global_ic("uniform", 100000, float(p.nf_physical_dim), 3)
    xv[:, 3:] = np.random.default_rng(8).normal(0, 0.4, (len(xv), 3)).astype(np.float32)
    g, o, og, oo = run_both(p, xv, pid, (0.3, 0.1, 0.1, 8.0), steps=2)
    assert og.np_total == oo.np_total == len(xv) and og.np_ghost == oo.np_ghost
    assert og.dt_f_acc == pytest.approx(oo.dt_f_acc, rel=1e-5) and og.dt_c_acc == pytest.approx(oo.dt_c_acc, rel=1e-5)
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 1e-4
        vin = np.stack([v0[q] for q in pg.tolist()])
        dg, do = xg[:, 3:].astype(np.float64) - vin, xo[:, 3:].astype(np.float64) - vin
        num += ((dg - do) ** 2).sum()
        den += (do ** 2).sum()
    assert observed("test_eight_logical_ranks_one_tile_each_like_the_bench_workload: kick, relative rms over both steps", np.sqrt(num / den), BAR_KICK_2STEP) <= BAR_KICK_2STEP


# ------------------------------------------------------------------ the drop-in itself: a Fortran MPI host
def _write_kernel_ascii(path, table):
    """kernels/wfxyz*.ascii as fine_kernel / coarse_kernel read them: '(3i4,3e16.8)' per (i,j,k), i fastest
    (kernel_initialization.f90:15-30,344-358); `table` is (n,n,n,3) indexed [k][j][i]."""
    n = table.shape[0]
    with open(path, "w") as f:
        for k in range(n):
            for j in range(n):
                for i in range(n):
                    f.write("%4d%4d%4d%16.8E%16.8E%16.8E\n" % ((i + 1, j + 1, k + 1) + tuple(float(v) for v in table[k, j, i])))


@pytest.mark.parametrize("cfg,kw", [("cfg1_8rank_pp", dict(ngp=True, ppint=True, pp_ext=True)), ("cfg1_8rank", dict(ngp=True)),
                                    ("cfg1_8rank_pencil", dict(ngp=True, lrckcorr=True, pencil=True))])
def test_fortran_mpi_host_calls_particle_mesh_through_the_adapter(tmp_path, cfg, kw):
    """`mpiexec -n 8 hip_mpi_driver`: the reference's COMMON blocks and its own mpi_initialize, `call particle_mesh`
    resolved by cubep3m_amd/fortran/particle_mesh_hip_mpi.f90 (ISO_C_BINDING + the three MPI transport callbacks),
    eight MPI ranks sharing this GPU -- against the oracle's eight simulated ranks.  The binary is built in the dev
    container from the reference's headers (oracle/build_ref.sh) and travels with the snapshot."""
    import os
    import shutil
    import subprocess

    from cubep3m_amd.group import split_global

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "oracle", "_ref", cfg)
    exe = os.path.join(d, "hip_mpi_driver")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("oracle/_ref/%s/hip_mpi_driver not built (dev container: bash oracle/build_ref.sh)" % cfg)
    os.makedirs(os.path.join(d, "kernels"), exist_ok=True)       # kernel_path = cubepm_root//'kernels/' (cubepm.par:10)
    _write_kernel_ascii(os.path.join(d, "kernels", "wfxyzf.3.ascii"), FINE_TABLE)
    _write_kernel_ascii(os.path.join(d, "kernels", "wfxyzc.2.ascii"), COARSE_TABLE)
    p = cfg1(nodes_dim=2, **kw)
    xv, pid = global_ic("clustered", 60000, float(p.nf_physical_dim), 314)
    parts = split_global(p, xv, pid, range(8))
    scal = np.asarray((0.01, 0.3, 0.3, 8.0), np.float32)
    nsteps = 2
    for r in range(8):
        x, q = parts[r]
        with open(tmp_path / ("in%d.bin" % r), "wb") as f:
            np.asarray([len(x), nsteps], np.int32).tofile(f)
            scal.tofile(f)
            np.ascontiguousarray(x, np.float32).tofile(f)
            np.ascontiguousarray(q, np.int64).tofile(f)
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([mpiexec, "-n", "8", exe, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    # the adapter is built with -DMPI_TIME: after every step it prints the GPU phase times the way the reference does (tag : max avg min
    # over the ranks, timers.f90:68-77), from p3m_hip_group_last_phase_ms
    import re

    try:   # keep the report of the PP build for INTEGRATION.md
        if cfg == "cfg1_8rank_pp":
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "mpi_time_report.txt"), "w") as f:
                f.write(res.stdout)
    except OSError:
        pass
    rows = re.findall(r"^\s*(pos updt|linklist|par pass|fm  mass|fm   fft|fm  kick|pp intra|pp   ext|cm  mass|cm force|cm   vel|del part)\s*:\s+(\S+)\s+(\S+)\s+(\S+)", res.stdout, re.M)
    assert len(rows) == 12 * nsteps, res.stdout[-3000:]
    for tag, mx, av, mn in rows:
        assert float(mx) >= float(av) >= float(mn) >= 0.0, (tag, mx, av, mn)
    assert all(float(mx) > 0 for tag, mx, _, _ in rows if tag in ("linklist", "fm   fft", "fm  kick", "cm force"))
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    for r in range(8):
        o.set_particles(r, *parts[r])
    for s in range(nsteps):
        oo = o.particle_mesh(float(scal[0]), float(scal[1]), float(scal[2]) if s == 0 else float(scal[1]), float(scal[3]))
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for r in range(8):
        raw = np.fromfile(tmp_path / ("out%d.bin" % r), np.uint8)
        n = int(raw[:4].view(np.int32)[0])
        dts = raw[4:20].view(np.float32)
        xg = raw[20:20 + 24 * n].view(np.float32).reshape(n, 6)
        pg = raw[20 + 24 * n:20 + 32 * n].view(np.int64)
        assert dts[0] == pytest.approx(oo.dt_f_acc, rel=1e-5) and dts[3] == pytest.approx(oo.dt_c_acc, rel=1e-5), r     # reduced over ranks
        if kw.get("pp_ext"):
            assert dts[1] == pytest.approx(oo.dt_pp_acc, rel=1e-5) and dts[2] == pytest.approx(oo.dt_pp_ext_acc, rel=1e-5), r
        xg, pg = by_pid(xg, pg)
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 1e-4
        vin = np.stack([v0[q] for q in pg.tolist()])
        dg, do = xg[:, 3:].astype(np.float64) - vin, xo[:, 3:].astype(np.float64) - vin
        num += ((dg - do) ** 2).sum()
        den += (do ** 2).sum()
    assert observed("test_fortran_mpi_host_calls_particle_mesh_through_the_adapter: kick, relative rms over both steps", np.sqrt(num / den), BAR_KICK_2STEP) <= BAR_KICK_2STEP


def test_standalone_run_reads_ic_and_writes_reference_checkpoints(tmp_path):
    """python -m cubep3m_amd.run: IC files in, the time loop, checkpoint files in the reference's format out; the
    checkpoint is read back and holds every particle once, inside the box, with the scale factor of the list."""
    from cubep3m_amd import io_formats as iof
    from cubep3m_amd import run as runmod
    from cubep3m_amd.group import split_global

    p = cfg1(nodes_dim=2, ngp=True)
    xv, pid = global_ic("uniform", 40000, float(p.nf_physical_dim), 17)
    xv[:, 3:] = np.random.default_rng(2).normal(0, 0.05, (len(xv), 3)).astype(np.float32)
    parts = split_global(p, xv, pid, range(8))
    ic, out = tmp_path / "ic", tmp_path / "out"
    ic.mkdir()
    for r in range(8):
        iof.write_ic(ic / ("xv%d.ic" % r), parts[r][0])
    rc = runmod.main(["--ic-dir", str(ic), "--out-dir", str(out), "--nodes-dim", "2", "--tiles", "2", "--nf-tile", "80", "--z-i", "49",
                      "--checkpoints", "48.0,47.5", "--projections", "47.8", "--max-nts", "60"])
    assert rc == 0
    n2 = p.nf_physical_dim
    masses = []
    for name in iof.projection_names(47.8):            # projection.f90's three files: a, then the nf_physical_dim^2 map
        a_p, m = iof.read_projection(out / name, n2)
        assert a_p == pytest.approx(1.0 / (1.0 + 47.8), rel=1e-4) and m.min() >= 0.0
        masses.append(float(m.astype(np.float64).sum()))
    mass_p = float(n2) ** 3 / len(xv)
    for mm in masses:                                   # a slab one rank thick: about half of the (uniform) mass
        assert 0.4 * mass_p * len(xv) < mm < 0.6 * mass_p * len(xv)
    total = 0
    for z in (48.0, 47.5):
        n_z = 0
        for r in range(8):
            nx, npid = iof.checkpoint_names(z, r)
            h, x = iof.read_checkpoint(out / nx)
            hp, q = iof.read_pid_checkpoint(out / npid)
            assert h.np_local == len(x) == len(q) and h.a == pytest.approx(1.0 / (1.0 + z), rel=1e-4)   # the step that lands on an output redshift is cut by timestep.f90 in f32
            # checkpoint.f90 writes right after the output step's half drift (cubepm.f90:176-185): a record may sit a drift length outside
            assert np.all((x[:, :3] >= -0.25) & (x[:, :3] < p.nf_physical_node_dim + 0.25))
            n_z += len(x)
        assert n_z == len(xv)
        total += n_z
    assert total == 2 * len(xv)


def test_bench_multi_process_launch_on_one_gpu():
    """bench.py for N > 1 (torch.distributed.run, one process per rank: the driver's launch line, which bench.py also issues itself when it
    is started without a launcher), here with two processes on this one GPU: the host side on gloo, RCCL's refusal of two ranks per device answered by the collective fall-back
    to the host transport.  Checks the launch contract (env, id broadcast, max-over-ranks timing, ONE JSON line)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the plain form `python bench.py --gpus 2 ...` (no launcher, WORLD_SIZE unset): bench.py starts torch.distributed.run itself, as a
    # child process, before it touches the GPU, and relays the one JSON line and the exit code
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--config", "cfg4_small", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["ranks_per_gpu"] == 4 and "roofline" in d and "cpu_baseline" not in d      # cpu_baseline: rank 0 at N = 1 only
    assert set(d["phase_ms"]) >= {"update_position", "link_list", "fine_fft", "fine_kick", "coarse_force"} and d["phase_ms"]["fine_fft"] > 0


def test_ghost_segments_grow_when_a_clustered_shell_overfills_them(monkeypatch):
    """The exchange segments start at a multiple of the uniform-density share of a face / edge / corner shell; a clustered
    shell that overfills one makes every process grow it (the counts of every rank are gathered first, so all processes
    take the same decision) and pack again -- the reference would have taken up to max_buf = 2.2 max_np floats per
    direction (cubepm.par:174).  Started here at 5 % of the uniform share so that every segment overflows."""
    monkeypatch.setenv("P3M_GHOST_SEG_FACTOR", "0.05")
    p = cfg1(nodes_dim=2, ngp=True, ppint=True, pp_ext=True)
    xv, pid = global_ic("clustered", 60000, float(p.nf_physical_dim), 123)
    g, o, og, oo = run_both(p, xv, pid, (0.01, 0.3, 0.3, 8.0), steps=2)
    assert og.np_total == oo.np_total == len(xv) and og.np_ghost == oo.np_ghost
    for name in ("dt_f_acc", "dt_c_acc", "dt_pp_acc", "dt_pp_ext_acc"):
        assert getattr(og, name) == pytest.approx(getattr(oo, name), rel=1e-5), name
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po) and np.abs(xg[:, :3] - xo[:, :3]).max() <= 1e-4
        assert rel_rms(xg[:, 3:], xo[:, 3:]) <= 1e-5


def test_capacity_overflow_in_the_ghost_pass_is_one_error_for_the_whole_group():
    """particle_pass.f90:136-139 (mpi_abort when np_local + np_buf > max_np) as an error code: with the counts of all ranks
    gathered, every process -- here every logical rank -- sees the same overflow and the step returns P3M_ECAPACITY
    before any payload moves."""
    from cubep3m_amd import lib
    from cubep3m_amd.group import ParticleMeshGroup

    p = cfg1(nodes_dim=2, ngp=True, density_buffer=1.0)
    g = ParticleMeshGroup(p, 0, 1, FINE_TABLE, COARSE_TABLE)
    cap = g.rank_context(0).derived(0)                    # max_np, cubepm.par:170-172
    # rank 0 nearly full, its neighbours' face shells packed: the arrivals no longer fit rank 0
    rng = np.random.default_rng(5)
    Nn = float(p.nf_physical_node_dim)
    for i, r in enumerate(g.local_ranks):
        n = cap - 10 if r == 0 else 30000
        xv = np.zeros((n, 6), np.float32)
        xv[:, :3] = rng.random((n, 3), dtype=np.float32) * np.float32(Nn)
        if r != 0:
            xv[:, :3] = xv[:, :3] * np.float32(0.3)          # near the low faces: ghosts of the neighbour at -x, -y, -z
        np.minimum(xv[:, :3], np.float32(Nn * (1 - 2e-6)), out=xv[:, :3])
        g.upload_particles(i, xv, np.arange(1, n + 1, dtype=np.int64) + r * 10 ** 7)
    with pytest.raises(lib.P3MError) as e:
        g.particle_mesh(0.01, 0.1, 0.0, 8.0)
    assert e.value.code == -3 and "max_np" in str(e.value)


def test_pid_slots_are_repacked_while_migrants_keep_arriving():
    """PIDs rest in pid_home; a migrant takes a new slot where it arrives and the slot it leaves is never freed, so the slots in
    use grow until pid_repack (particles.hip) compacts them -- when the holes exceed a quarter of the records.  Fast particles in
    small ranks (a fifth of a rank's records change owner per step) make that happen every other step: six steps, then PIDs,
    positions and velocities by PID against the oracle's eight ranks (delete_particles.f90:17-47, particle_pass.f90:69-722)."""
    p = cfg1(nodes_dim=2, ngp=True)
    box = float(p.nf_physical_dim)
    xv, pid = global_ic("uniform", 60000, box, 31)
    xv[:, 3:] = np.random.default_rng(32).normal(0, 25.0, (len(xv), 3)).astype(np.float32)     # |v| dt ~ 5 cells of a 64-cell rank
    g, o, og, oo = run_both(p, xv, pid, (0.5, 0.2, 0.2, 8.0), steps=6)
    assert og.np_total == oo.np_total == len(xv) and og.np_ghost == oo.np_ghost
    moved = 0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 2e-3 and rel_rms(xg[:, 3:], xo[:, 3:]) <= 1e-5
        moved += len(pg)
    assert moved == len(xv)


@pytest.mark.parametrize("switch", ["P3M_COARSE_PER_RANK", "P3M_COARSE_COPY", "P3M_ONE_STREAM"])
def test_per_rank_coarse_path_stays_at_parity(switch):
    """P3M_ONE_STREAM=1 keeps the coarse force on the main stream (no second stream, no events).  P3M_COARSE_PER_RANK=1 runs the distributed coarse transform rank by rank (the path of pencil decompositions and of mesh
    sizes without register-stage FFT kernels) where the default batches every stage over the local ranks; P3M_COARSE_COPY=1
    keeps the batched stages but moves the three redistributions as messages (what several processes run) where a single
    process gathers / stores them in place: both held to the same tests, in a child process (the switches are read when the
    group is created)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    # (the whole-step comparisons reuse the oracle results the parent's tests left in the oracle cache, tests/oracle_lib.py)
    sel = {"P3M_ONE_STREAM": "(eight_logical_ranks_match_oracle and (uniform-kw0 or clustered-kw1)) or distributed_coarse_mesh_vs_oracle",
           "P3M_COARSE_PER_RANK": "distributed_coarse_mesh_vs_oracle or (eight_logical_ranks_match_oracle and uniform) or nc256",
           "P3M_COARSE_COPY": "distributed_coarse_mesh_vs_oracle or (eight_logical_ranks_match_oracle and uniform) or nc256 or (distributed_coarse_mesh_at_the_bench_size and False)"}[switch]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_group.py"), os.path.join(here, "test_gpu_baseline_sizes.py"),
                        os.path.join(here, "test_gpu_slab1024.py"), "-q", "-x", "-m", "gpu", "-k", sel],
                       env=dict(os.environ, **{switch: "1"}), cwd=os.path.dirname(here), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("kind,kw", [("uniform", dict(ngp=True)), ("clustered", dict(ngp=True, ppint=True, pp_ext=True))])
def test_a_stream_per_local_rank_stays_at_parity(kind, kw, monkeypatch):
    """P3M_GROUP_STREAMS=8 (read when the group is created): between the ghost pass and the step's host wait every local rank queues on a
    stream of its own -- the fork behind the ghost pass, the coarse force waiting for every rank's coarse density, every rank's kick
    waiting for the coarse force, the join before the host wait (group.hip).  Same kernels, same results: held to the oracle by the
    eight-rank test itself (whose oracle results are in the cache by now)."""
    monkeypatch.setenv("P3M_GROUP_STREAMS", "8")
    test_eight_logical_ranks_match_oracle(kind, kw, False)


def test_group_takes_the_kernels_a_host_already_holds():
    """p3m_hip_group_set_kernels_raw: kern_f and every rank's z-slab of kern_c in the reference's layout (what kernel_checkpoint.f90
    writes) instead of the two ascii tables -- redistributed once into the transposed k-space order; the step then matches the
    oracle like the table-built group does."""
    from cubep3m_amd.group import ParticleMeshGroup

    p = cfg1(nodes_dim=2, ngp=True, ppint=True, pp_ext=True, lrckcorr=True)
    xv, pid = global_ic("clustered", 60000, float(p.nf_physical_dim), 42)
    o = ol.Oracle(p)
    o.set_kernel_tables(FINE_TABLE, COARSE_TABLE)
    kf, kc = o.kern_f(), o.kern_c()                       # (nf, nf, nf/2+1, 3), (nc, nc, nc/2+1, 3) = [kz][ky][kx][comp]
    s = p.nc_slab
    g = ParticleMeshGroup(p, 0, 1, set_kernels=False)
    g.set_kernels_raw(kf, [kc[r * s:(r + 1) * s] for r in g.local_ranks])
    parts = g.scatter_global(xv, pid)
    for r in range(8):
        o.set_particles(r, *parts[r])
    og, oo = g.particle_mesh(0.01, 0.3, 0.3, 8.0), o.particle_mesh(0.01, 0.3, 0.3, 8.0)
    for name in ("dt_f_acc", "dt_c_acc", "dt_pp_acc", "dt_pp_ext_acc"):
        assert getattr(og, name) == pytest.approx(getattr(oo, name), rel=1e-5), name
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po)
        vin = np.stack([v0[q] for q in pg.tolist()])
        num += ((xg[:, 3:].astype(np.float64) - xo[:, 3:].astype(np.float64)) ** 2).sum()
        den += ((xo[:, 3:].astype(np.float64) - vin) ** 2).sum()
    assert np.sqrt(num / den) <= 1e-5


def test_five_steps_with_every_exchange_on_rccl_and_the_second_stream():
    """One RCCL communicator serves two streams: the ghost pass and the all-reduces on the main stream, the coarse transform's
    exchanges on the second one (underneath the fine-mesh sweeps).  Five steps with every exchange forced through
    ncclSend/ncclRecv (P3M_ONE_STREAM unset), particles crossing rank boundaries, against the oracle."""
    import os

    assert os.environ.get("P3M_ONE_STREAM", "0") != "1"
    p = cfg1(nodes_dim=2, ngp=True, ppint=True, pp_ext=True)
    xv, pid = global_ic("clustered", 60000, float(p.nf_physical_dim), 123)
    g, o, og, oo = run_both(p, xv, pid, (0.01, 0.3, 0.3, 8.0), force_rccl=True, steps=5)
    info = g.comm_info()
    assert info["comm_count"] == 1 and info["comm_rank"] == 0 and len(info["uuid"]) == 32
    assert og.np_total == oo.np_total == len(xv) and og.np_ghost == oo.np_ghost
    for name in ("dt_f_acc", "dt_c_acc", "dt_pp_acc", "dt_pp_ext_acc"):
        assert observed("test_five_steps_with_every_exchange_on_rccl_and_the_second_stream: " + name, abs(getattr(og, name) / getattr(oo, name) - 1.0), BAR_DT_5STEP) <= BAR_DT_5STEP, name
    v0 = dict(zip(pid.tolist(), xv[:, 3:]))
    num = den = 0.0
    for i, r in enumerate(g.local_ranks):
        xg, pg = by_pid(*g.download_particles(i))
        xo, po = by_pid(*o.get_particles(r))
        assert np.array_equal(pg, po), "rank %d holds a different particle set" % r
        assert np.abs(xg[:, :3] - xo[:, :3]).max() <= 2e-4
        vin = np.stack([v0[q] for q in pg.tolist()])
        num += ((xg[:, 3:].astype(np.float64) - xo[:, 3:].astype(np.float64)) ** 2).sum()
        den += ((xo[:, 3:].astype(np.float64) - vin) ** 2).sum()
    assert observed("test_five_steps_with_every_exchange_on_rccl_and_the_second_stream: velocity change, relative rms over five steps", np.sqrt(num / den), BAR_KICK_5STEP) <= BAR_KICK_5STEP


def test_one_rank_per_gpu_share_of_config5_fits_the_device():
    """ranks_per_gpu = 1 at BASELINE config 5's size: ONE logical rank of the 2x2x2 decomposition (1024^3 fine cells, 2^3 tiles
    of 560, room for 1.3 x 512^3 particles, PPINT + PP_EXT) as process 0 of 8 -- its coarse slab buffers are those of a
    512^3 mesh in 8 slabs (nc_slab = 64), its ghost segments are opened to the reference's max_buf (2.2 max_np floats per
    direction, cubepm.par:174: a face holds 0.37 max_np records).  Everything is allocated at creation; it has to fit the
    288 GB of one MI355X with room to spare."""
    import os

    import torch

    from cubep3m_amd.group import ParticleMeshGroup
    from cubep3m_amd.params import Params

    p = Params(nodes_dim=2, tiles_node_dim=2, nf_tile=560, ngp=True, ppint=True, pp_ext=True, density_buffer=1.3, cores=8)
    assert p.nc_dim == 512 and p.nc_slab == 64
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    old = os.environ.get("P3M_GHOST_SEG_FACTOR")
    os.environ["P3M_GHOST_SEG_FACTOR"] = "%.3f" % (2.2 / 6.0 * p.nf_physical_node_dim / p.nf_buf)
    try:
        g = ParticleMeshGroup(p, 0, 8, set_kernels=False)
    finally:
        if old is None:
            del os.environ["P3M_GHOST_SEG_FACTOR"]
        else:
            os.environ["P3M_GHOST_SEG_FACTOR"] = old
    assert g.nlocal == 1 and g.local_ranks == [0]
    free1, _ = torch.cuda.mem_get_info()
    used = free0 - free1
    g.close()
    assert total >= 280e9                              # an MI355X
    assert 40e9 < used < 0.6 * total, used              # measured: see DESIGN (Multi-GPU)
    print("config-5 share, one rank per GPU: %.1f GB of %.1f GB" % (used / 1e9, total / 1e9))
