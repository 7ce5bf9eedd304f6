"""TEST INFRASTRUCTURE (dev container only): runs the REFERENCE'S OWN timestep.f90 object code (oracle/_ref/<cfg>/libref.so,
built where it lies by oracle/build_ref.sh) over the scenario of tests/time_scenarios.py and saves what it produced.
usage: python ref_time_run.py <cfg> <scenario> <out.npz>   (child process of tests/golden/make_ref_timestep.py)"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_lib  # noqa: E402
from time_scenarios import SCENARIOS, expansion_grid, limits_of_step  # noqa: E402

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def main():
    cfg, name, out = sys.argv[1:4]
    sc = SCENARIOS[name]
    L = C.CDLL(ref_lib.so_path(cfg))
    L.ref_init()
    L.ref_time_set.argtypes = [f32p, i32p, f32p, C.c_int, f32p, C.c_int, f32p, C.c_int]
    L.ref_time_get.argtypes = [f32p, i32p]
    L.ref_time_params.argtypes = [f32p]
    L.ref_expansion.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    par = np.zeros(8, np.float32)
    L.ref_time_params(par)
    chk, prj, hf = (np.asarray(sc[k], np.float32) for k in ("a_checkpoint", "a_projection", "a_halofind"))
    a, tau, t, dt, dt_old = sc["a0"], sc["tau0"], 0.0, 0.0, 0.0
    nts, cur = 0, [1, 1, 1]
    rows_r, rows_i = [], []
    for k in range(sc["steps"]):
        lim = limits_of_step(name, k)
        rv = np.asarray([a, tau, t, dt, dt_old] + list(lim), np.float32)
        iv = np.asarray([nts] + cur, np.int32)
        L.ref_time_set(rv, iv, chk, len(chk), prj, len(prj), hf, len(hf))
        L.ref_timestep()
        ro, io = np.zeros(8, np.float32), np.zeros(5, np.int32)
        L.ref_time_get(ro, io)
        rows_r.append(ro.copy()); rows_i.append(io.copy())
        a, _, _, dt, dt_old, _, tau, t = (np.float32(v) for v in ro)
        nts = int(io[0])
        # what checkpoint.f90 / projection.f90 / halofind.f90 and cubepm.f90:217 do after an output step
        if io[1]: cur[0] += 1
        if io[2]: cur[1] += 1
        if io[3]: cur[2] += 1
        if io[1] or io[2] or io[3]: dt = np.float32(0.0)
        if io[4] or a > 1.0: break
    grid = expansion_grid()
    ex = np.zeros((len(grid), 2), np.float32)
    for i, (a0, dt0) in enumerate(grid):
        d1, d2 = C.c_float(), C.c_float()
        L.ref_expansion(C.c_float(a0), C.c_float(dt0), C.byref(d1), C.byref(d2))
        ex[i] = (d1.value, d2.value)
    np.savez(out, params=par, out_r=np.asarray(rows_r), out_i=np.asarray(rows_i), expansion=ex)
    L.ref_finalize()


if __name__ == "__main__":
    main()
