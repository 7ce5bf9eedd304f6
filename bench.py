#!/usr/bin/env python
"""Benchmark of the P3M gravity step (`particle_mesh`) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4|cfg2|cfg3|...] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one full `particle_mesh` call (drift -> ghost pass + cell sort -> per-tile fine PM [+PP]
-> coarse PM with the slab FFT -> ghost removal) on synthetic uniform particles that are resident in
HBM when the timed region starts.  Default workload = BASELINE.json configs[3]: 1024^3 fine mesh /
512^3 particles in the reference's 2x2x2 cubic decomposition (8 logical ranks); with N GPUs each
process drives one GPU and owns 8/N logical ranks (strong scaling; exchanges between ranks on one GPU
are device copies, between GPUs RCCL send/recv over xGMI).
Prints ONE JSON line: metric particle-updates/s (whole job), `roofline` for the dominant kernel (an FFT
line pass) measured live with HIP events on the library's stream, and, at N=1, `cpu_baseline` (the CPU
oracle -- a C port of the reference path -- timed on a bounded sample on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# this pool's host driver supports dmabuf IPC only: without it RCCL between processes fails in hipIpcGetMemHandle
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from cubep3m_amd.kernels import default_tables  # noqa: E402
from cubep3m_amd.params import Params  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    # BASELINE.json configs[3]: 1024^3 fine / 512^3 particles, 2x2x2 ranks, one 560^3 tile per rank, 256^3 coarse slab FFT
    "cfg4": dict(params=dict(nodes_dim=2, tiles_node_dim=1, nf_tile=560, ngp=True, density_buffer=1.3), nside_rank=256,
                 workload="1024^3 fine mesh / 512^3 particles, PM-only (NGP), 2x2x2 logical ranks, nf_tile=560 (one tile per rank), "
                          "256^3 coarse mesh with slab FFT + all-to-all transpose"),
    # configs[0]: 64^3 fine / 32^3 particles, PM-only
    "cfg1": dict(params=dict(tiles_node_dim=2, nf_tile=80, ngp=True), nside_rank=32,
                 workload="64^3 fine mesh / 32^3 particles, PM-only (NGP), nf_tile=80, 2^3 tiles"),
    # configs[1]: 256^3 / 128^3, PM-only
    "cfg2": dict(params=dict(tiles_node_dim=2, nf_tile=176, ngp=True, density_buffer=1.5), nside_rank=128,
                 workload="256^3 fine mesh / 128^3 particles, PM-only (NGP), nf_tile=176, 2^3 tiles, 64^3 coarse"),
    "cfg2_t4": dict(params=dict(tiles_node_dim=4, nf_tile=112, ngp=True, density_buffer=1.5), nside_rank=128,
                    workload="256^3 fine mesh / 128^3 particles, PM-only (NGP), nf_tile=112, 4^3 tiles, 64^3 coarse"),
    # configs[2]: 256^3 / 128^3, PM+PP+extended PP
    "cfg3": dict(params=dict(tiles_node_dim=2, nf_tile=176, ngp=True, ppint=True, pp_ext=True, density_buffer=1.5), nside_rank=128,
                 workload="256^3 fine mesh / 128^3 particles, PM+PP+PP_EXT, nf_tile=176, 2^3 tiles"),
    # configs[1] with the CIC mass assignment / interpolation (the reference built without -DNGP)
    "cfg2_cic": dict(params=dict(tiles_node_dim=2, nf_tile=176, ngp=False, density_buffer=1.5), nside_rank=128,
                     workload="256^3 fine mesh / 128^3 particles, PM-only (CIC), nf_tile=176, 2^3 tiles, 64^3 coarse"),
    # configs[3]'s problem with the short-range forces on (the P3M step of the north star; configs[4]'s flags at configs[3]'s size)
    "cfg4_pp": dict(params=dict(nodes_dim=2, tiles_node_dim=1, nf_tile=560, ngp=True, ppint=True, pp_ext=True, density_buffer=1.3), nside_rank=256,
                    workload="1024^3 fine mesh / 512^3 particles, PM+PP+PP_EXT (NGP), 2x2x2 logical ranks, nf_tile=560 (one tile per rank), "
                             "256^3 coarse mesh with slab FFT + all-to-all transpose"),
    # configs[3] with the CIC mass assignment / interpolation on the fine mesh (fine_cic_mass.f90:13-43, particle_mesh_threaded.f90:289-316):
    # the reference built without -DNGP
    "cfg4_cic": dict(params=dict(nodes_dim=2, tiles_node_dim=1, nf_tile=560, ngp=False, density_buffer=1.3), nside_rank=256,
                     workload="1024^3 fine mesh / 512^3 particles, PM-only (CIC fine mesh), 2x2x2 logical ranks, nf_tile=560 (one tile per rank), "
                              "256^3 coarse mesh with slab FFT + all-to-all transpose"),
    # configs[3]'s problem as ONE rank of 2^3 tiles (what a single-GPU user of the reference would build: nodes_dim = 1)
    "cfg4_1rank": dict(params=dict(nodes_dim=1, tiles_node_dim=2, nf_tile=560, ngp=True, density_buffer=1.3), nside_rank=512,
                       workload="1024^3 fine mesh / 512^3 particles, PM-only (NGP), one rank of 2^3 tiles of nf_tile=560, 256^3 coarse mesh"),
    # one rank's share of configs[3] on its own
    "big512": dict(params=dict(tiles_node_dim=1, nf_tile=560, ngp=True, density_buffer=1.3), nside_rank=256,
                   workload="512^3 fine mesh / 256^3 particles (one rank's share of 1024^3/512^3), PM-only, nf_tile=560, 1 tile"),
    # small multi-rank problem for quick checks
    "cfg4_small": dict(params=dict(nodes_dim=2, tiles_node_dim=1, nf_tile=112, ngp=True, density_buffer=1.5), nside_rank=32,
                       workload="128^3 fine mesh / 64^3 particles, PM-only, 2x2x2 logical ranks, nf_tile=112"),
}


def make_particles(nside, box, seed=12345):
    n = nside ** 3
    rng = np.random.default_rng(seed)
    xv = np.zeros((n, 6), np.float32)
    xv[:, :3] = rng.random((n, 3), dtype=np.float32) * np.float32(box)
    np.minimum(xv[:, :3], np.float32(box * (1 - 2e-6)), out=xv[:, :3])
    xv[:, 3:] = rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    return xv


FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (vector), spec
# flop per pair evaluation as the kernels spend them (sub 3, r^2 5, rsqrt 1, magnitudes 5, force 6, accumulate 3 = 23; the
# extended kernel adds the polynomial taper: q 1, q^3 2, q^5 2, taper 4, scaling 3 = 35)
PP_FLOP_INTRA, PP_FLOP_EXT = 23.0, 35.0


def clustered(nside, box, seed, frac, nblobs, sigma):
    n = nside ** 3
    rng = np.random.default_rng(seed)
    nb = int(frac * n)
    pos = rng.random((n, 3)) * box
    centers = rng.random((nblobs, 3)) * box
    pos[:nb] = centers[rng.integers(0, nblobs, nb)] + rng.normal(0, sigma, (nb, 3))
    pos = np.mod(pos, box).astype(np.float32)
    np.minimum(pos, np.float32(box * (1 - 2e-6)), out=pos)
    xv = np.zeros((n, 6), np.float32)
    xv[:, :3] = pos
    return xv


def pp_leg():
    """The two short-range kernels on their own (BASELINE configs[2] geometry: 256^3 cells / 128^3 particles, PPINT + PP_EXT):
    pair evaluations per launch, evaluations/s and the fraction of the FP32 vector peak they amount to, for the uniform IC and
    for the clustered IC of SURVEY Appendix C (30 % of the particles in Gaussian blobs of sigma 0.6 cells, default_rng(2024))
    at two blob sizes."""
    from cubep3m_amd.particle_mesh import ParticleMesh

    p = Params(**CONFIGS["cfg3"]["params"])
    fine, coarse = default_tables()
    out = {"geometry": CONFIGS["cfg3"]["workload"], "peak_fp32_valu_TFLOPs": FP32_VALU_PEAK_TFLOPS,
           "flop_per_evaluation": {"intra": PP_FLOP_INTRA, "extended": PP_FLOP_EXT},
           "note": "an evaluation is one partner summed into one kicked record (a pair of two kicked records is evaluated twice)"}
    def rates(ms_i, ms_e, n_i, n_e):
        return {"intra": {"evaluations": n_i, "ms": ms_i, "evaluations_per_s": n_i / (ms_i * 1e-3) if ms_i > 0 else None,
                          "valu_frac": n_i * PP_FLOP_INTRA / (ms_i * 1e-3) / (FP32_VALU_PEAK_TFLOPS * 1e12) if ms_i > 0 else None},
                "extended": {"evaluations": n_e, "ms": ms_e, "evaluations_per_s": n_e / (ms_e * 1e-3) if ms_e > 0 else None,
                             "valu_frac": n_e * PP_FLOP_EXT / (ms_e * 1e-3) / (FP32_VALU_PEAK_TFLOPS * 1e12) if ms_e > 0 else None}}

    ics = {"uniform": lambda: make_particles(128, 256.0),
           "clustered_205_per_blob": lambda: clustered(128, 256.0, 2024, 0.3, 3072, 0.6),     # Appendix C's blobs at Appendix C's density
           "clustered_13k_per_blob": lambda: clustered(128, 256.0, 2024, 0.3, 48, 0.6)}      # the same 48 blobs holding 64x the particles
    for name, gen in ics.items():
        g = ParticleMesh(p, fine, coarse)
        g.upload_particles(gen())
        g.link_list_and_pass()
        ms_i, ms_e, n_i, n_e = g.time_pp(0.5, 0.0, 8.0, reps=5)       # dt = 0: the repeated kicks leave the velocities alone
        g.delete_particles()
        g.close()
        out[name] = rates(ms_i, ms_e, n_i, n_e)
    # The same kernels on the HEADLINE's tile (one rank of the default workload: a 560 tile, 256^3 particles), in the order every step but the
    # first after an upload finds them in: one whole step first, so that the velocities are reached in the last step's sorted order
    # (tests/ppbench.py ... big steady).  These are the figures the pm_pp and clustered.pm_pp legs are made of.
    p5 = Params(tiles_node_dim=1, nf_tile=560, ngp=True, ppint=True, pp_ext=True, density_buffer=1.3)
    out["tile560_steady"] = {"geometry": "one 560 tile (512^3 cells / 256^3 particles: one rank of the headline), velocities in steady-state order"}
    for name, gen in {"uniform": lambda: make_particles(256, 512.0), "clustered_205_per_blob": lambda: clustered(256, 512.0, 2024, 0.3, 3072 * 8, 0.6)}.items():
        g = ParticleMesh(p5, fine, coarse)
        g.upload_particles(gen())
        g.particle_mesh(0.5, 0.0, 0.0, 8.0)
        g.update_position(0.0, 0.0)
        g.link_list_and_pass()
        ms_i, ms_e, n_i, n_e = g.time_pp(0.5, 0.0, 8.0, reps=5)
        g.delete_particles()
        g.close()
        out["tile560_steady"][name] = rates(ms_i, ms_e, n_i, n_e)
    return out


def cpu_baseline(scal):
    """The CPU oracle (tests/oracle_lib.py: a C port of the reference path with the reference's own OpenMP regions -- the
    tile loop of particle_mesh_threaded.f90:84, coarse_mass.f90:83, coarse_velocity.f90:137, update_position.f90:68,
    coarse_force.f90:37-86; link_list, particle_pass and delete_particles are serial there and here) on a bounded sample: a
    256^3-cell sub-volume (BASELINE configs[1]: 128^3 particles of the same uniform density), full particle_mesh steps, 4^3
    tiles of 112 (64 work items; 2^3 tiles of 176 on boxes with <= 8 cores).  BASELINE.md section 3: one warm-up step, then the
    thread count is swept (8 ... the box's cores, one timed step each), then THREE timed steps at the best count with the phases
    timed one by one -- that is `value` -- and the same three steps for the PM + PPINT + PP_EXT flag set (`pm_pp`, the figure that
    stands beside the bench line's pm_pp leg).  Checker-side code, used here only as the reported baseline."""
    import ctypes

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    cores = os.cpu_count() or 1
    try:
        cores_avail = len(os.sched_getaffinity(0))
    except Exception:
        cores_avail = cores
    key = "cfg2" if cores_avail <= 8 else "cfg2_t4"
    fine, coarse = default_tables()
    os.environ["OMP_NUM_THREADS"] = str(max(1, cores_avail))
    omp = None
    for name in ("libgomp.so.1", "libomp.so", "libomp.so.5"):
        try:
            omp = ctypes.CDLL(name)
            break
        except OSError:
            continue
    xv = make_particles(128, 256.0)
    a_mid, dt, dt_old, mass_p = scal
    phases = ("update_position", "link_list", "particle_pass", "fine_mesh", "coarse_mesh", "delete_particles")
    t_all = time.perf_counter()

    def timed_steps(o, nsteps):
        split = dict.fromkeys(phases, 0.0)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            for name, call in (("update_position", lambda: o.update_position(dt, dt_old)), ("link_list", o.link_list),
                               ("particle_pass", o.particle_pass), ("fine_mesh", lambda: o.fine_mesh(a_mid, dt, mass_p)),
                               ("coarse_mesh", lambda: o.coarse_mesh(a_mid, dt, mass_p)), ("delete_particles", o.delete_particles)):
                t1 = time.perf_counter()
                call()
                split[name] += time.perf_counter() - t1
        el = time.perf_counter() - t0
        return {"particle_updates_per_s": len(xv) * nsteps / el, "s_per_step": el / nsteps, "steps": nsteps,
                "phase_s_per_step": {k: v / nsteps for k, v in split.items()}}

    def oracle_for(**flags):
        q = Params(**dict(CONFIGS[key]["params"], **flags))
        o = ol._RawOracle(q)   # (ol.Oracle memoises whole steps on disk: the timed object is the plain one)
        o.set_kernel_tables(fine, coarse)
        o.set_particles(0, xv)
        return q, o

    counts = sorted({t for t in (8, 16, 32, 64, 128, 256) if t <= cores_avail} | {max(1, min(cores_avail, 256))})
    p, o = oracle_for()
    ntile = p.tiles_node_dim ** 3
    if omp is not None:
        omp.omp_set_num_threads(min(32, cores_avail))
    o.particle_mesh(a_mid, dt, dt_old, mass_p)              # warm-up step (first touch of the tile work spaces)
    sweep = {}
    for threads in counts:
        if omp is not None:
            omp.omp_set_num_threads(threads)
        sweep[threads] = timed_steps(o, 1)
        if time.perf_counter() - t_all > 20.0:
            break
    best = max(sweep, key=lambda t: sweep[t]["particle_updates_per_s"])
    if omp is not None:
        omp.omp_set_num_threads(best)
    final = timed_steps(o, 3)
    o.close()
    # the short-range flag set at the same thread count (the tile loop is the only OpenMP region that grows)
    _, opp = oracle_for(ppint=True, pp_ext=True)
    opp.particle_mesh(a_mid, dt, dt_old, mass_p)
    pp = timed_steps(opp, 3 if time.perf_counter() - t_all < 28.0 else 1)
    opp.close()
    el_all = time.perf_counter() - t_all
    return {"value": final["particle_updates_per_s"], "unit": "particle-updates/s", "cores": best, "kind": "port",
            "host_cpus": cores, "host_cpus_usable": cores_avail, "timed_steps": final["steps"], "s_per_step": final["s_per_step"],
            "phase_s_per_step": final["phase_s_per_step"], "thread_sweep": sweep,
            "pm_pp": {"value": pp["particle_updates_per_s"], "unit": "particle-updates/s", "cores": best, "timed_steps": pp["steps"],
                      "s_per_step": pp["s_per_step"], "phase_s_per_step": pp["phase_s_per_step"],
                      "flags": "PM + PPINT + PP_EXT (the flag set of the bench line's pm_pp leg), same sample"},
            "sample": "full particle_mesh steps of a 256^3-cell / 128^3-particle sub-volume (same density, nf_tile=%d, %d^3 tiles = %d work items) "
                      "on the CPU oracle (C port of the reference path with the reference's OpenMP regions; link_list, particle_pass, "
                      "delete_particles serial as in the reference): one warm-up step, one timed step at each of %s threads, then %d timed "
                      "steps at the best count (%d threads: %.2f s per step; %s) and %d timed steps with PPINT + PP_EXT (%.2f s per step) on a "
                      "box with %d CPUs, %.1f s in all.  Context (survey-time probe of the "
                      "reference Fortran itself, amdflang + MKL FFT shim, dev container, 4 OpenMP threads, same 256^3/128^3 problem, "
                      "BASELINE.md section 2): 1.65e6 particle-updates/s PM-only, 0.96e6 with PP + extended PP; the reference's own 2007 log "
                      "(8 cores, 128^3 particles, PM+PP): 8.8e4"
                      % (p.nf_tile, p.tiles_node_dim, ntile, sorted(sweep), final["steps"], best, final["s_per_step"],
                         ", ".join("%s %.3f" % (k, v) for k, v in final["phase_s_per_step"].items()), pp["steps"], pp["s_per_step"], cores, el_all)}


def open_group(p, torch, dist, rank, world, ddev, uid, require_rccl, dist_backend):
    """The group of logical ranks this process drives, connected by RCCL when there is more than one process.  If the RCCL
    communicator cannot be set up (agreed on by all ranks) the run exits non-zero -- unless --allow-host-transport was
    given: then the same exchanges go through the host-callback transport over a gloo group and the line is marked."""
    from cubep3m_amd.group import ParticleMeshGroup

    if world == 1:
        return ParticleMeshGroup(p, rank, world, set_kernels=False), "none"
    from cubep3m_amd.group import torch_transport
    from cubep3m_amd.lib import P3MError

    ok, grp = 1, None
    try:
        grp = ParticleMeshGroup(p, rank, world, unique_id=uid, set_kernels=False)
    except P3MError as e:
        ok = 0
        print("rank %d: RCCL transport unavailable (%s)" % (rank, e), file=sys.stderr, flush=True)
    flag = torch.tensor([ok], device=ddev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return grp, "rccl"
    if require_rccl:
        raise SystemExit("RCCL transport unavailable (a multi-GPU line needs it; --allow-host-transport falls back to gloo)")
    if grp is not None:
        grp.close()
    gloo = dist.new_group(backend="gloo") if dist_backend == "nccl" else None
    return ParticleMeshGroup(p, rank, world, set_kernels=False, transport=torch_transport(dist, gloo)), "host callbacks over gloo (RCCL unavailable)"


def comm_audit(grp, dist, rank, world):
    """Per-rank record of the device and the RCCL communicator this process really uses (printed to stderr by every rank,
    gathered into the JSON line by rank 0): device UUID, ncclCommCount, ncclCommUserRank."""
    info = dict(grp.comm_info(), rank=rank, logical_ranks=list(grp.local_ranks))
    print("bench rank %d: device %d uuid %s ncclCommCount %d ncclCommUserRank %d logical ranks %s"
          % (rank, info["device"], info["uuid"], info["comm_count"], info["comm_rank"], info["logical_ranks"]), file=sys.stderr, flush=True)
    if world == 1:
        return [info]
    allinfo = [None] * world
    dist.all_gather_object(allinfo, info)
    return allinfo


def slab_leg(args, torch, dist, rank, world, local_dev, ddev, require_rccl, embedded=False):
    """--config slab1024: BASELINE configs[3] read literally (SURVEY section 8, config note (ii)) -- a 1024^3 real coarse field,
    slab-decomposed over the eight logical ranks of the 2x2x2 decomposition (nc_slab = 128, fftw3ds.f90:103-183), one forward
    and three inverse transforms per step (coarse_force.f90:18-90).  A "step" is one coarse_force on device-resident density:
    cube -> slab redistribution, forward x / y passes, all-to-all transpose, z pass, multiply, three inverse transforms with
    their transposes, slab -> cube, force halo."""
    from cubep3m_amd.group import rccl_unique_id

    p = Params(nodes_dim=2, tiles_node_dim=4, nf_tile=560, coarse_only=True, cores=1, lrckcorr=True, device=local_dev)
    assert p.nc_dim == 1024 and p.nc_slab == 128
    uid = None
    if world > 1:
        t = torch.zeros(128, dtype=torch.uint8, device=ddev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(t, 0)
        uid = bytes(t.cpu().numpy().tobytes())
    grp, transport = open_group(p, torch, dist, rank, world, ddev, uid, require_rccl, args.dist_backend)
    _, coarse = default_tables()
    grp.set_kernel_tables(None, coarse)
    audit = comm_audit(grp, dist, rank, world)
    n = p.nc_node_dim
    for i, r in enumerate(grp.local_ranks):     # a sparse random density (1/8 particle per fine cell = 8 per coarse cell on average)
        rng = np.random.default_rng(4000 + r)
        grp.set_coarse_density(i, rng.poisson(8.0, (n, n, n)).astype(np.float32) * 8.0)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        grp.coarse_transform("force")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        grp.coarse_transform("force")           # ends with a stream synchronisation
    sync()
    el = time.perf_counter() - t0
    ms_fwd = grp.coarse_transform("forward", reps=5)
    ms_force = grp.coarse_transform("force", reps=5)
    if dist is not None:
        tt = torch.tensor([el, ms_fwd, ms_force], dtype=torch.float64, device=ddev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el, ms_fwd, ms_force = (float(v) for v in tt.tolist())
    if rank == 0:
        nc = p.nc_dim
        ms_step = 1e3 * el / args.steps
        slab_bytes = 4.0 * (nc + 2) * nc * p.nc_slab                  # one rank's slab, SURVEY 8a row a15
        alg = 10.5 * slab_bytes * p.nodes                             # SURVEY 8d "coarse FFT 10.5 x 4(nc+2) nc nc_slab" per rank
        per_peer = grp.coarse_exchange_bytes
        res = {"metric": "coarse_slab_fft_transforms_per_sec", "value": 4.0 * args.steps / el, "unit": "1024^3 transforms/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "transport_fallback": transport.startswith("host callbacks"),
               "config": {"workload": "literal 1024^3 real coarse field, slab-decomposed over 2x2x2 = 8 logical ranks (nc_slab = 128): one forward + "
                                      "three inverse transforms with the K_c multiply, cube<->slab redistribution and force halo (coarse_force.f90)",
                          "name": "slab1024", "logical_ranks": p.nodes, "ranks_per_gpu": p.nodes // world, "transport": transport},
               "ms_per_transform": ms_step / 4.0,
               "events": {"forward_ms": ms_fwd, "coarse_force_ms": ms_force, "force_minus_forward_ms": ms_force - ms_fwd},
               # the HBM side of the transforms: algorithmic bytes per GPU against the HBM peak, whatever the number of GPUs (with
               # several GPUs the all-to-all below travels over xGMI on top of it: xgmi_link gives that side against a link's rate)
               "roofline": {"bound": "hbm", "kernel": "coarse_force (1 forward + 3 inverse 1024^3 transforms, all passes and exchanges)",
                            "achieved": alg / (ms_force * 1e-3) / 1e9 / world, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                            "frac": alg / (ms_force * 1e-3) / 1e9 / world / HBM_PEAK_GBS, "traffic": None,
                            "algorithmic_bytes": alg,
                            "xgmi_link": None if world == 1 else {
                                "bytes_per_link_per_step": 4 * per_peer * (p.nodes // world) ** 2, "peak_GBs": 153.0,
                                "achieved_GBs": 4 * per_peer * (p.nodes // world) ** 2 / (ms_force * 1e-3) / 1e9,
                                "frac": 4 * per_peer * (p.nodes // world) ** 2 / (ms_force * 1e-3) / 1e9 / 153.0,
                                "note": "what one GPU sends to ONE other GPU in the four transposes of a step (every pair of their logical ranks "
                                        "exchanges per_peer bytes per transpose) over the step's whole time: a lower bound of the link's rate"}},
               "all_to_all": {"bytes_per_peer_per_transpose": per_peer, "peers": p.nodes - 1, "transposes_per_step": 4,
                              "bytes_per_rank_per_step": 4 * per_peer * (p.nodes - 1),
                              "note": "each rank sends nc_slab x (nc/2+1 padded to 16) x nc_slab complex to every other rank per transform "
                                      "(SURVEY 8d: 67 MB per link at G = 8); between ranks of one GPU these are device copies"},
               "ranks": audit}
        if embedded:        # a leg of the default line (one GPU): the dictionary goes into the headline's JSON
            grp.close()
            res.pop("ranks")
            return res
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
    grp.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS) + ["slab1024"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the non-headline legs of the default run (PM+PP step at the headline's size, PP kernel rates)")
    ap.add_argument("--require-rccl", action="store_true", help="(default for --gpus > 1; kept for old command lines)")
    ap.add_argument("--allow-host-transport", action="store_true",
                    help="multi-GPU runs: fall back to the host-callback transport over gloo when RCCL cannot be set up, instead of "
                         "exiting non-zero (such a line says nothing about xGMI and is marked transport_fallback)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the host side (gloo: debugging on fewer GPUs than ranks)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one per GPU) as a CHILD process group -- before anything here has
        # touched the GPU, and without exec (a process that initialised the GPU must not be replaced) -- and hand its result through
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    local_dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    ddev = "cuda" if args.dist_backend == "nccl" else "cpu"      # where the host side's small collectives live
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend="gloo")

    from cubep3m_amd.group import ParticleMeshGroup, rccl_unique_id

    require_rccl = world > 1 and not args.allow_host_transport and args.dist_backend == "nccl"
    if world > 1:
        # a multi-GPU line must come from `world` DIFFERENT devices: gather every rank's device UUID
        props = torch.cuda.get_device_properties(local_dev)
        my = "%s/%d" % (getattr(props, "uuid", None) or "dev", local_dev)
        ids = [None] * world
        dist.all_gather_object(ids, my)
        if len(set(ids)) < world and args.dist_backend != "gloo":
            raise SystemExit("bench.py --gpus %d sees only %d distinct device(s) %s: one process per GPU is required "
                             "(--dist-backend gloo runs several ranks on one device for debugging)" % (world, len(set(ids)), sorted(set(ids))))
    if args.config == "slab1024":
        return slab_leg(args, torch, dist, rank, world, local_dev, ddev, require_rccl)

    cfg = CONFIGS[args.config]
    p = Params(**cfg["params"])
    p.device = local_dev
    if p.nodes % world:
        raise SystemExit("%d logical ranks cannot be split over %d GPUs" % (p.nodes, world))
    uid = None
    if world > 1:   # rank 0 creates the RCCL id, torch.distributed (RCCL) broadcasts it
        t = torch.zeros(128, dtype=torch.uint8, device=ddev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(t, 0)
        uid = bytes(t.cpu().numpy().tobytes())

    fine, coarse = default_tables()
    grp, transport = open_group(p, torch, dist, rank, world, ddev, uid, require_rccl, args.dist_backend)
    grp.set_kernel_tables(fine, coarse)
    audit = comm_audit(grp, dist, rank, world)
    box = float(p.nf_physical_node_dim)
    nside = cfg["nside_rank"]
    mass_p = float((p.nf_physical_node_dim / nside) ** 3)  # fine cells per particle = 8
    scal = (0.5, 0.05, 0.05, mass_p)                       # late-time scalar set of SURVEY section 8d
    a_mid, dt, dt_old, _ = scal
    n_local = 0
    for i, r in enumerate(grp.local_ranks):               # every logical rank gets its own uniform cube
        xv = make_particles(nside, box, seed=12345 + r)
        grp.upload_particles(i, xv, np.arange(1, len(xv) + 1, dtype=np.int64) + r * len(xv))
        n_local += len(xv)
        del xv
    n_total = nside ** 3 * p.nodes

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        grp.particle_mesh(a_mid, dt, dt_old, mass_p)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = grp.particle_mesh(a_mid, dt, dt_old, mass_p)   # ends with a stream sync: the dt limits are returned to the host
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=ddev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    assert out.np_total == n_total, (out.np_total, n_total)
    value = n_total * args.steps / el
    # per-phase GPU times of one more step (through the C ABI: what a reference host built with -DMPI_TIME prints, timers.f90:68-77);
    # a collective step: every rank takes it, rank 0 reports the spans of its own ranks
    grp.phase_timing(True)
    grp.particle_mesh(a_mid, dt, dt_old, mass_p)
    phase_ms = grp.last_phase_ms()
    grp.phase_timing(False)

    if rank == 0:
        # ---- roofline of the dominant kernel, measured live with HIP events on the library's stream (rank 0's first context)
        pm = grp.rank_context(0)
        S = 4.0 * (p.nf_tile + 2) * p.nf_tile ** 2              # bytes of one fine real/complex array
        passes = {}
        nb = 1
        for i, name in enumerate(pm.FFT_PASSES):
            ms, nb = pm.time_fft_pass(i, reps=10)
            passes[name] = ms
        # per sweep over the batch, one launch each: x_fwd, y_fwd, z_fwd, z_inv_fused (3 components), y_inv (3), x_inv_extract (3)
        # on the step's path: x_fwd, y_fwd, z_inv_fused (forward z pass + multiply + inverse z pass of all three components in
        # one kernel), y_inv, x_inv_extract; the stand-alone z_fwd is only used when the Green's functions are built
        # what NGP whole steps run instead of x_inv_extract + the kick: the inverse x pass with the kick inside (kick_fused.hip)
        fused_ms = None
        if p.ngp:
            try:
                fused_ms, _ = pm.time_fft_pass(7, reps=10)
                passes["x_inv_kick_fused"] = fused_ms
            except Exception:
                fused_ms = None
        dom = max(("x_fwd", "y_fwd", "z_inv_fused", "y_inv", "x_inv_extract"), key=lambda k: passes[k])
        sweep_ms = pm.time_fine_sweep(mass_p, reps=3)
        ntile = p.tiles_node_dim ** 3
        # the forward 3-D transform of one tile is 2*S algorithmic bytes (one read + one write), one force
        # component is 2.5*S (read rho-hat, read half-size kernel, write).  Three axis passes per transform: a forward pass
        # launch over `nb` tiles carries (2/3)*S*nb, an inverse pass launch (all three components) 3*(2.5/3)*S*nb.
        alg_bytes = {"x_fwd": 2.0 / 3.0, "y_fwd": 2.0 / 3.0, "y_inv": 2.5, "x_inv_extract": 2.5,
                     "z_inv_fused": 2.5 + 2.0 / 3.0}[dom] * S * nb   # the fused kernel carries the forward z pass too
        achieved = alg_bytes / (passes[dom] * 1e-3) / 1e9
        traffic = None
        kname = None
        sweep_traffic = None
        in_step_traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                from make_traffic import fft_source_sha16

                recj = json.load(open(tf)).get(args.config, {})
                # byte counts recorded for other FFT sources than the ones running now are stale: traffic stays null
                allp = recj.get("passes", {}) if recj.get("fft_source_sha16") == fft_source_sha16() else {}
                rec = allp.get(dom, {})
                traffic = rec.get("hbm_bytes_per_launch")
                kname = rec.get("kernel")
                step_passes = ("x_fwd", "y_fwd", "z_inv_fused", "y_inv", "x_inv_extract")
                if all(k in allp for k in step_passes):   # the five launches of one sweep: PMC bytes and their own launch times
                    sweep_traffic = sum(allp[k]["hbm_bytes_per_launch"] for k in step_passes)
                in_step = ("x_fwd", "y_fwd", "z_inv_fused", "y_inv", "x_inv_kick_fused")
                if all(k in allp for k in in_step):       # the step's own five launches (the last one carries the kick's bytes too)
                    in_step_traffic = sum(allp[k]["hbm_bytes_per_launch"] for k in in_step)
            except Exception:
                traffic = None
        if not kname:
            kname = {"x_fwd": "k_fft_x_fwd*", "y_fwd": "k_fft_lines*", "z_inv_fused": "k_fft_lines3*", "y_inv": "k_fft_lines*", "x_inv_extract": "k_fft_x_inv*"}[dom]
        roofline = {"bound": "hbm", "kernel": "%s (%s pass, %d tile(s)/launch, nf_tile=%d)" % (kname, dom, nb, p.nf_tile), "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    # the byte counts are NOT measured in this run: they are replayed from the PMC passes recorded under profiles/ (guarded by a
                    # hash of the FFT sources: stale records give null)
                    "traffic_source": ("replayed profiles/pmc_traffic.json@" + str(recj.get("fft_source_sha16"))) if traffic else None,
                    "ms_per_launch": passes[dom], "pass_ms": passes,
                    # the same launch against the bytes it really moved (PMC): what the memory system sees
                    "traffic_GBs": (traffic / (passes[dom] * 1e-3) / 1e9) if traffic else None,
                    "traffic_frac": (traffic / (passes[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    # The step's own sweep: the NGP deposit rides on the sort (no deposit kernel) and the last launch is the fused inverse-x +
                    # kick pass, which carries the kick of particle_mesh_threaded.f90:208-270 on top of the transform: four FFT launches + that
                    # one against the same 10.5 S.  "fine_sweep" below is the stand-alone sweep (k_ngp_counts + five launches + a force box).
                    # Timed AS THE STEP RUNS IT: the spans of one more whole step's own phase timers -- fine_fft (the four FFT
                    # launches of every rank) + fine_kick (the fused pass with its velocity stores, the survivor count and k_kick_fix), with the
                    # coarse transform on the second stream underneath -- divided by the ranks on this GPU.  "standalone_dry_ms" is the sum of the
                    # stand-alone pass timings, whose last launch runs WITHOUT the velocity stores and the fix-up: a lower bound, not the step's figure.
                    "fine_sweep_in_step": None if fused_ms is None else {
                        "ms": (phase_ms["fine_fft"] + phase_ms["fine_kick"]) / len(grp.local_ranks),
                        "source": "phase_ms of a whole step (GPU event spans), per rank",
                        "algorithmic_bytes": 10.5 * S * ntile,
                        "frac": 10.5 * S * ntile / ((phase_ms["fine_fft"] + phase_ms["fine_kick"]) / len(grp.local_ranks) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "standalone_dry_ms": passes["x_fwd"] + passes["y_fwd"] + passes["z_inv_fused"] + passes["y_inv"] + fused_ms,
                        "traffic": in_step_traffic,
                        "what": "x_fwd, y_fwd, fused z, y_inv, inverse x with the NGP kick, the coarse kick and the survivor count inside (one rank's tiles)"},
                    "fine_sweep": {"ms": sweep_ms, "algorithmic_bytes": 10.5 * S * ntile,
                                   "achieved_GBs": 10.5 * S * ntile / (sweep_ms * 1e-3) / 1e9,
                                   "frac": 10.5 * S * ntile / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   # the five FFT launches against the HBM bytes rocprofv3's counters report for them
                                   "fft_passes_ms": sum(passes[k] for k in ("x_fwd", "y_fwd", "z_inv_fused", "y_inv", "x_inv_extract")),
                                   "fft_traffic": sweep_traffic,
                                   "fft_traffic_GBs": (sweep_traffic / (sum(passes[k] for k in ("x_fwd", "y_fwd", "z_inv_fused", "y_inv", "x_inv_extract")) * 1e-3) / 1e9) if sweep_traffic else None,
                                   "fft_traffic_frac": (sweep_traffic / (sum(passes[k] for k in ("x_fwd", "y_fwd", "z_inv_fused", "y_inv", "x_inv_extract")) * 1e-3) / 1e9 / HBM_PEAK_GBS) if sweep_traffic else None}}
        res = {
            "metric": "particle_updates_per_sec", "value": value, "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # true: the exchanges between processes did NOT go over RCCL/xGMI but through host callbacks over gloo -- such a
            # line is no statement about xGMI scaling
            "transport_fallback": transport.startswith("host callbacks"),
            "config": {"workload": cfg["workload"], "name": args.config, "particles": n_total, "logical_ranks": p.nodes,
                       "ranks_per_gpu": p.nodes // world, "tiles_per_rank": ntile, "nf_tile": p.nf_tile, "transport": transport,
                       "flags": {"ngp": p.ngp, "ppint": p.ppint, "pp_ext": p.pp_ext}},
            "roofline": roofline,
            "ranks": audit,
        }
        res["phase_ms"] = phase_ms
        if world == 1 and args.config == "cfg4" and not args.no_extra:
            # non-headline legs (rank 0, one GPU): the same 512^3-particle problem with PPINT + PP_EXT on, and the two
            # short-range kernels on their own
            grp.close()

            def side_leg(cfg_name, gen, data, env=None):
                """full steps of another flag set / another IC on the headline's geometry, timed like the headline (env: library switches,
                read when the group is created)"""
                p2 = Params(**CONFIGS[cfg_name]["params"])
                p2.device = local_dev
                saved = {k: os.environ.get(k) for k in (env or {})}
                os.environ.update(env or {})
                try:
                    g2 = ParticleMeshGroup(p2, 0, 1, fine, coarse)
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                for i, r in enumerate(g2.local_ranks):
                    xv = gen(r)
                    g2.upload_particles(i, xv, np.arange(1, len(xv) + 1, dtype=np.int64) + r * len(xv))
                    del xv
                for _ in range(2):
                    g2.particle_mesh(a_mid, dt, dt_old, mass_p)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                k2 = max(3, min(args.steps, 6))
                for _ in range(k2):
                    o2 = g2.particle_mesh(a_mid, dt, dt_old, mass_p)
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t1
                assert o2.np_total == n_total
                leg = {"metric": "particle_updates_per_sec", "value": n_total * k2 / el2, "ms_per_step": 1e3 * el2 / k2, "steps": k2,
                       "workload": CONFIGS[cfg_name]["workload"], "data": data}
                g2.phase_timing(True)
                g2.particle_mesh(a_mid, dt, dt_old, mass_p)
                leg["phase_ms"] = g2.last_phase_ms()
                g2.phase_timing(False)
                if not p2.ngp:
                    # the north star's "fine-mesh FFT+CIC sweep": CIC deposit + the five FFT launches of one tile against 10.5 S, and
                    # the CIC gather (maximum + interpolation + kick in one pass over the force box) against its own bytes
                    pm2 = g2.rank_context(0)
                    S2 = 4.0 * (p2.nf_tile + 2) * p2.nf_tile ** 2
                    nt2 = p2.tiles_node_dim ** 3
                    sw = pm2.time_fine_sweep(mass_p, reps=3)
                    ga = pm2.time_fine_gather(reps=3)
                    fbx = p2.nf_physical_tile_dim + 3
                    gbytes = 12.0 * fbx ** 3 * nt2 + 48.0 * (n_total / p2.nodes)       # 12 fb^3 per tile + 48 B per particle
                    leg["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "fine_sweep": {"ms": sw, "algorithmic_bytes": 10.5 * S2 * nt2, "achieved_GBs": 10.5 * S2 * nt2 / (sw * 1e-3) / 1e9,
                                                      "frac": 10.5 * S2 * nt2 / (sw * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                      "what": "CIC deposit (fine_cic_mass.f90:13-43) + forward x, y, fused z, inverse y, inverse x of one rank's tiles"},
                                       "cic_gather": {"ms": ga, "algorithmic_bytes": gbytes, "achieved_GBs": gbytes / (ga * 1e-3) / 1e9,
                                                      "frac": gbytes / (ga * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                      # in the step the same pass also carries the coarse kick (coarse_velocity.f90:137-179) and the coarse
                                                      # transform runs underneath: the step's own span of it, per rank
                                                      "in_step_ms": leg["phase_ms"]["fine_kick"] / len(g2.local_ranks),
                                                      "what": "max |F|^2 + CIC interpolation + kick (particle_mesh_threaded.f90:208-223,289-316), one pass over the force box; stand-alone, WITHOUT the coarse kick that rides on it in the step (in_step_ms)"},
                                       "sweep_and_gather": {"ms": sw + ga, "algorithmic_bytes": 10.5 * S2 * nt2 + gbytes,
                                                            "frac": (10.5 * S2 * nt2 + gbytes) / ((sw + ga) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                       # the same bytes against the step's own spans (fine_mass + fine_fft + fine_kick, per rank)
                                       "sweep_and_gather_in_step": {
                                           "ms": (leg["phase_ms"]["fine_mass"] + leg["phase_ms"]["fine_fft"] + leg["phase_ms"]["fine_kick"]) / len(g2.local_ranks),
                                           "frac": (10.5 * S2 * nt2 + gbytes) / ((leg["phase_ms"]["fine_mass"] + leg["phase_ms"]["fine_fft"] + leg["phase_ms"]["fine_kick"]) / len(g2.local_ranks) * 1e-3) / 1e9 / HBM_PEAK_GBS}}
                g2.close()
                if p2.ppint:
                    leg.update(dt_pp_acc=o2.dt_pp_acc, dt_pp_ext_acc=o2.dt_pp_ext_acc)
                return leg

            def uniform_ic(r):
                return make_particles(nside, box, seed=12345 + r)

            def clustered_ic(r):   # SURVEY Appendix C's recipe at its density: 30 % of the particles in blobs of ~205, sigma 0.6 cells
                xv = clustered(nside, box, 2024 + r, 0.3, 48 * (nside // 32) ** 3, 0.6)
                xv[:, 3:] = np.random.default_rng(99 + r).normal(0, 0.05, (len(xv), 3)).astype(np.float32)
                return xv

            def moving_ic(r):      # the headline's particles with velocities that carry them 0.3 cells per step and axis (sigma_v dt = 6 x 0.05)
                xv = make_particles(nside, box, seed=12345 + r)
                xv[:, 3:] = np.random.default_rng(777 + r).normal(0, 6.0, (len(xv), 3)).astype(np.float32)
                return xv

            res["cic"] = side_leg("cfg4_cic", uniform_ic, "synthetic uniform (the headline's particles)")
            # the headline's timed steps move nobody (velocities of 0.05 cells per unit time: the sort sees sorted input); a production step
            # displaces particles by a fraction of a cell (update_position.f90:68-76 with timestep.f90:106-115's dt)
            mdata = "synthetic uniform positions, Gaussian velocities with sigma_v dt = 0.3 cells per step and axis"
            res["moving"] = {"pm": side_leg("cfg4", moving_ic, mdata), "pm_pp": side_leg("cfg4_pp", moving_ic, mdata)}
            res["pm_pp"] = side_leg("cfg4_pp", uniform_ic, "synthetic uniform (the headline's particles)")
            # the headline's step with a stream per logical rank (P3M_GROUP_STREAMS, opt-in: group.hip): launch gaps, kernel tails and the
            # latency-bound particle kernels of one rank run underneath another rank's passes.  Not the headline: with kernels of different
            # ranks sharing the device a kernel's duration is no longer its own, and the roofline above is per kernel
            res["rank_streams"] = dict(side_leg("cfg4", uniform_ic, "synthetic uniform (the headline's particles)", env={"P3M_GROUP_STREAMS": "8"}),
                                       what="the headline's workload with P3M_GROUP_STREAMS=8: one stream per logical rank between the ghost pass and the step's host wait")
            # the same two step types on a clustered particle set: dense cells, unequal rows, heavy pair lists
            cdata = "synthetic clustered: 30 % of the particles in Gaussian blobs of ~205 particles, sigma 0.6 cells (SURVEY Appendix C's recipe at its density)"
            res["clustered"] = {"pm": side_leg("cfg4", clustered_ic, cdata), "pm_pp": side_leg("cfg4_pp", clustered_ic, cdata)}
            res["pp"] = pp_leg()
            # BASELINE configs[3] read literally: the stand-alone 1024^3 slab transform (--config slab1024), three coarse_force calls
            sa = argparse.Namespace(**dict(vars(args), steps=3, warmup=1))
            res["slab1024"] = slab_leg(sa, torch, None, 0, 1, local_dev, ddev, False, embedded=True)
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(scal)
            if "pm_pp" in res:   # the CPU figure of the same flag set beside the PM + PP leg
                res["pm_pp"]["cpu_baseline"] = {k: res["cpu_baseline"]["pm_pp"][k] for k in ("value", "unit", "cores", "timed_steps", "s_per_step")}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        grp.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
