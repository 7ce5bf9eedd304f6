#!/usr/bin/env python
"""Benchmark of the P3M gravity step (`particle_mesh`) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg1|big512] [--no-cpu]

A "step" is one full `particle_mesh` call (drift -> ghost pass + cell sort -> per-tile fine PM
[+PP] -> coarse PM -> ghost removal) on synthetic uniform particles that are resident in HBM when
the timed region starts.  Prints ONE JSON line (see the task contract): metric particle-updates/s,
plus `roofline` for the dominant kernel (the strided FFT line pass) measured live with HIP events on
the library's stream, and `cpu_baseline` (the CPU oracle, a port of the reference path, timed on the
same workload on this box's host cores).
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

from cubep3m_amd.kernels import default_tables  # noqa: E402
from cubep3m_amd.params import Params  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    # BASELINE.json configs[0]: 64^3 fine / 32^3 particles, PM-only
    "cfg1": dict(params=dict(tiles_node_dim=2, nf_tile=80, ngp=True), nside=32,
                 workload="64^3 fine mesh / 32^3 particles, PM-only (NGP), nf_tile=80, 2^3 tiles"),
    # configs[1]: 256^3 / 128^3, PM-only
    "cfg2": dict(params=dict(tiles_node_dim=2, nf_tile=176, ngp=True, density_buffer=1.5), nside=128,
                 workload="256^3 fine mesh / 128^3 particles, PM-only (NGP), nf_tile=176, 2^3 tiles, 64^3 coarse"),
    "cfg2_t4": dict(params=dict(tiles_node_dim=4, nf_tile=112, ngp=True, density_buffer=1.5), nside=128,
                    workload="256^3 fine mesh / 128^3 particles, PM-only (NGP), nf_tile=112, 4^3 tiles, 64^3 coarse"),
    # configs[2]: 256^3 / 128^3, PM+PP+extended PP
    "cfg3": dict(params=dict(tiles_node_dim=2, nf_tile=176, ngp=True, ppint=True, pp_ext=True, density_buffer=1.5), nside=128,
                 workload="256^3 fine mesh / 128^3 particles, PM+PP+PP_EXT, nf_tile=176, 2^3 tiles"),
    # one GPU's share of configs[3] (1024^3 fine / 512^3 particles on 8 GPUs): 512^3 fine, 256^3 particles
    "big512": dict(params=dict(tiles_node_dim=1, nf_tile=560, ngp=True, density_buffer=1.3), nside=256,
                   workload="512^3 fine mesh / 256^3 particles (one GPU's share of 1024^3/512^3), PM-only, nf_tile=560, 1 tile"),
}


def make_particles(nside, box, seed=12345):
    n = nside ** 3
    rng = np.random.default_rng(seed)
    xv = np.zeros((n, 6), np.float32)
    xv[:, :3] = rng.random((n, 3), dtype=np.float32) * np.float32(box)
    np.minimum(xv[:, :3], np.float32(box * (1 - 2e-6)), out=xv[:, :3])
    xv[:, 3:] = rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    return xv


def cpu_baseline(p: Params, xv, scal, budget_s=30.0):
    """Times the CPU oracle (tests/oracle_lib.py; a C port of the reference path, OpenMP over tiles like the
    reference's `!$omp do`) on the same workload.  Checker-side code, used here only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    fine, coarse = default_tables()
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, p.tiles_node_dim ** 3))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = ol.Oracle(p)
    o.set_kernel_tables(fine, coarse)
    o.set_particles(0, xv)
    a_mid, dt, dt_old, mass_p = scal
    t0 = time.perf_counter()
    steps = 0
    while True:
        o.particle_mesh(a_mid, dt, dt_old, mass_p)
        steps += 1
        if time.perf_counter() - t0 > 0.4 * budget_s or steps >= 3:
            break
    el = time.perf_counter() - t0
    return {"value": len(xv) * steps / el, "unit": "particle-updates/s", "cores": threads, "kind": "port",
            "sample": "%d full particle_mesh step(s) of the same workload on the CPU oracle (C port of the reference path, "
                      "OpenMP over fine tiles as the reference does), %.1f s" % (steps, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if world > 1:
        raise SystemExit("multi-rank bench not available in this revision (see DESIGN.md 'Multi-GPU')")
    torch.cuda.set_device(local_rank)

    from cubep3m_amd.particle_mesh import ParticleMesh

    cfg = CONFIGS[args.config]
    p = Params(**cfg["params"])
    p.device = local_rank
    fine, coarse = default_tables()
    box = float(p.nf_physical_node_dim)
    xv = make_particles(cfg["nside"], box)
    n = len(xv)
    mass_p = float((p.nf_physical_node_dim / cfg["nside"]) ** 3)  # (fine cells)/np = 8
    scal = (0.5, 0.05, 0.05, mass_p)  # late-time scalar set of SURVEY section 8d
    a_mid, dt, dt_old, _ = scal

    pm = ParticleMesh(p, fine, coarse)
    pm.upload_particles(xv)           # inputs resident in HBM before the timed region
    for _ in range(args.warmup):
        pm.particle_mesh(a_mid, dt, dt_old, mass_p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pm.particle_mesh(a_mid, dt, dt_old, mass_p)   # each call ends with a stream sync (dt limits are returned)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert out.np_total == n, (out.np_total, n)
    value = n * args.steps / el

    # ---- roofline of the dominant kernel, measured live with HIP events on the library's stream
    S = 4.0 * (p.nf_tile + 2) * p.nf_tile ** 2                  # bytes of one fine real/complex array
    passes = {}
    for i, name in enumerate(pm.FFT_PASSES):
        ms, nb = pm.time_fft_pass(i, reps=20)
        passes[name] = ms
    # per sweep over the batch, one launch each: x_fwd, y_fwd, z_fwd, z_inv_fused (3 components), y_inv (3), x_inv_extract (3)
    dom = max(("y_fwd", "z_fwd", "z_inv_fused", "y_inv"), key=lambda k: passes[k])
    sweep_ms = pm.time_fine_sweep(mass_p, reps=5)
    ntile = p.tiles_node_dim ** 3
    # the forward 3-D transform of one tile is 2*S algorithmic bytes (one read + one write), one force
    # component is 2.5*S (read rho-hat, read half-size kernel, write).  This implementation spends three axis passes on a
    # transform, so a forward pass launch over `nb` tiles carries (2/3)*S*nb and an inverse pass launch (all three
    # components in one launch) 3*(2.5/3)*S*nb.
    alg_bytes = ((2.0 / 3.0) if dom.endswith("fwd") else 2.5) * S * nb
    achieved = alg_bytes / (passes[dom] * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(args.config, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_fft_lines (%s pass, %d tiles/launch)" % (dom, nb), "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "ms_per_launch": passes[dom], "pass_ms": passes,
                "fine_sweep": {"ms": sweep_ms, "algorithmic_bytes": 10.5 * S * ntile,
                               "achieved_GBs": 10.5 * S * ntile / (sweep_ms * 1e-3) / 1e9,
                               "frac": 10.5 * S * ntile / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}

    res = {
        "metric": "particle_updates_per_sec", "value": value, "unit": "particle-updates/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"], "name": args.config, "particles": n, "tiles": ntile, "nf_tile": p.nf_tile,
                   "flags": {"ngp": p.ngp, "ppint": p.ppint, "pp_ext": p.pp_ext}},
        "roofline": roofline,
    }
    if not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(p, xv, scal)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
