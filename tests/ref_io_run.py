"""TEST INFRASTRUCTURE (dev container only): lets the REFERENCE'S OWN particle_initialization.o read an IC file written by
cubep3m_amd.io_formats and its checkpoint.o write checkpoint files (oracle/_ref/<cfg>/libref.so).
usage: python ref_io_run.py <cfg> <out.npz>      (child process of tests/golden/make_ref_io.py)"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_lib  # noqa: E402
from cubep3m_amd import io_formats as iof  # noqa: E402

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")

# state the checkpoint is written from
RV = np.asarray([0.3125, 7.25, -3.5, 0.43, 0.021, 1.75, 8.0, 2.2, 0.25, -0.5, 1.125], np.float32)   # a,t,tau,dt_f,dt_pp,dt_c,mass_p,z,shake(3)
IV = np.asarray([137, 3, 2, 4], np.int32)                                                           # nts, cur_checkpoint, cur_projection, cur_halofind


def ic_particles(n=157, box=64.0, seed=5):
    rng = np.random.default_rng(seed)
    xv = np.empty((n, 6), np.float32)
    xv[:, :3] = rng.random((n, 3), dtype=np.float32) * np.float32(box)
    xv[:, 3:] = rng.normal(0, 0.7, (n, 3)).astype(np.float32)
    return xv


def main():
    cfg, out = sys.argv[1:3]
    d = os.path.dirname(ref_lib.so_path(cfg))          # ic_path = output_path = this directory (oracle/build_ref.sh)
    xv = ic_particles()
    ic = os.path.join(d, "xv0.ic")
    iof.write_ic(ic, xv, binary=False)
    ref = ref_lib.Ref(cfg)
    L = ref.L
    L.ref_particle_initialize()                         # reads xv0.ic (and, with -DPID_FLAG, writes PID0.ic)
    xr, pr = ref.get_particles()
    L.ref_checkpoint.argtypes = [f32p, i32p]
    L.ref_checkpoint(RV, IV)
    xvn, pidn = iof.checkpoint_names(float(RV[7]), 0)
    res = dict(ic_xv=xv, ic_bytes=np.fromfile(ic, np.uint8), ref_read_xv=xr, ref_pid=pr, rv=RV, iv=IV,
               ckpt_xv_bytes=np.fromfile(os.path.join(d, xvn), np.uint8), ckpt_name=np.asarray(xvn))
    if os.path.exists(os.path.join(d, pidn)):
        res["ckpt_pid_bytes"] = np.fromfile(os.path.join(d, pidn), np.uint8)
    if os.path.exists(os.path.join(d, "PID0.ic")):
        res["pid_ic_bytes"] = np.fromfile(os.path.join(d, "PID0.ic"), np.uint8)
    np.savez(out, **res)
    L.ref_finalize()


if __name__ == "__main__":
    main()
