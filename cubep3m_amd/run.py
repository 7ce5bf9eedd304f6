"""A small stand-alone run in the reference's terms: read `xv<rank>.ic` (particle_initialization.f90:296-332), loop
timestep -> particle_mesh -> output steps (cubepm.f90:103-236), write `<z>xv<rank>.dat` / `<z>PID<rank>.dat`
(checkpoint.f90) -- the pieces of cubep3m_amd.timestep, cubep3m_amd.io_formats and the HIP gravity step put together.
Single process; nodes_dim^3 logical ranks share the GPU.

    python -m cubep3m_amd.run --ic-dir IC --out-dir OUT --nf-tile 80 --tiles 2 --z-i 50 --checkpoints 20,10 --ppint --pp-ext
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import io_formats as iof
from .group import ParticleMeshGroup
from .params import Params
from .timestep import Simulation, TimeParams, new_state


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ic-dir", required=True)
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--nodes-dim", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--nf-tile", type=int, default=80)
    ap.add_argument("--z-i", type=float, required=True, help="initial redshift (parameters: z_i)")
    ap.add_argument("--checkpoints", required=True, help="comma separated redshifts, descending (input/checkpoints)")
    ap.add_argument("--projections", default="", help="comma separated redshifts of the density projections (input/projections)")
    ap.add_argument("--omega-m", type=float, default=0.24)
    ap.add_argument("--omega-l", type=float, default=0.76)
    ap.add_argument("--ppint", action="store_true")
    ap.add_argument("--pp-ext", action="store_true")
    ap.add_argument("--cic", action="store_true", help="CIC instead of NGP on the fine mesh")
    ap.add_argument("--binary", action="store_true", help="-DBINARY file layout")
    ap.add_argument("--lrckcorr", action="store_true", help="-DLRCKCORR coarse kernel correction")
    ap.add_argument("--coarse-ngp", action="store_true", help="-DCOARSE_NGP coarse deposit and gather")
    ap.add_argument("--pencil", action="store_true", help="coarse FFT in 2-D pencils (the p3dfft build) instead of slabs; nodes_dim > 1")
    ap.add_argument("--max-nts", type=int, default=4000)
    a = ap.parse_args(argv)

    p = Params(nodes_dim=a.nodes_dim, tiles_node_dim=a.tiles, nf_tile=a.nf_tile, ngp=not a.cic, ppint=a.ppint, pp_ext=a.pp_ext,
               lrckcorr=a.lrckcorr, coarse_ngp=a.coarse_ngp, pencil=a.pencil and a.nodes_dim > 1)
    g = ParticleMeshGroup(p, 0, 1)
    npart = 0
    for i, r in enumerate(g.local_ranks):
        xv = iof.read_ic(os.path.join(a.ic_dir, "xv%d.ic" % r), binary=a.binary)
        # PID(i) = i + rank*np_local (particle_initialization.f90:334-343, assumes equal np_local)
        g.upload_particles(i, xv, np.arange(1, len(xv) + 1, dtype=np.int64) + np.int64(r) * len(xv))
        npart += len(xv)
    mass_p = float(p.nf_physical_dim) ** 3 / npart            # particle_initialization.f90:382
    zs = [float(z) for z in a.checkpoints.split(",")]
    zp = [float(z) for z in a.projections.split(",") if z]
    tp = TimeParams(omega_m=a.omega_m, omega_l=a.omega_l, a_checkpoint=[1.0 / (1.0 + z) for z in zs], a_projection=[1.0 / (1.0 + z) for z in zp])
    st = new_state(1.0 / (1.0 + a.z_i))
    os.makedirs(a.out_dir, exist_ok=True)

    def on_projection(sim):   # cubepm.f90:189-204: link_list, particle_pass, projection; projection.f90:56-113 writes the files
        s = sim.st
        z = zp[s.cur_projection - 1]
        pxy, pxz, pyz, tot = g.projection(mass_p)
        for name, m in zip(iof.projection_names(z), (pxy, pxz, pyz)):
            iof.write_projection(os.path.join(a.out_dir, name), s.a, m, binary=a.binary)
        print("projection z=%.3f  a=%.6f  total projected mass=%.6g" % (z, s.a, tot), flush=True)

    def on_output(sim):
        s = sim.st
        if s.checkpoint_step:
            on_checkpoint(sim)
        if s.projection_step:
            on_projection(sim)

    def on_checkpoint(sim):
        s = sim.st
        z = zs[s.cur_checkpoint - 1]
        for i, r in enumerate(g.local_ranks):
            xv, pid = g.download_particles(i)
            h = iof.P3MCkptHeader()
            h.a, h.t, h.tau, h.nts = s.a, s.t, s.tau, s.nts
            h.dt_f_acc, h.dt_pp_acc, h.dt_c_acc, h.mass_p = sim.last.dt_f_acc, sim.last.dt_pp_acc, sim.last.dt_c_acc, mass_p
            h.cur_checkpoint, h.cur_projection, h.cur_halofind = s.cur_checkpoint + 1, s.cur_projection, s.cur_halofind   # checkpoint.f90:52
            nx, npid = iof.checkpoint_names(z, r)
            iof.write_checkpoint(os.path.join(a.out_dir, nx), h, xv, binary=a.binary, ppint=a.ppint)
            iof.write_pid_checkpoint(os.path.join(a.out_dir, npid), h, pid, binary=a.binary, ppint=a.ppint)
        print("checkpoint z=%.3f  a=%.6f  nts=%d" % (z, s.a, s.nts), flush=True)

    sim = Simulation(g, tp, st, mass_p, max_nts=a.max_nts, on_output=on_output)
    while sim.step():
        if st.nts % 20 == 0:
            print("nts=%d a=%.6f dt=%.5f" % (st.nts, st.a, st.dt), flush=True)
    print("finished: nts=%d a=%.6f" % (st.nts, st.a), flush=True)
    g.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
