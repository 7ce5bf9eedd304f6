"""Measurement tool: the Fortran drop-in (`call particle_mesh` from an MPI host linked with the reference's own
mpi_initialization.o and COMMON blocks, oracle/hip_mpi_driver.f90 + cubep3m_amd/fortran/particle_mesh_hip_mpi.f90) timed
per step with and without resident particles.  8 MPI ranks of 256^3 cells / 128^3 particles each = 512^3 fine mesh,
256^3 particles; on a one-GPU box all ranks share the GPU and talk through the host transport (MPI).
    python3 tests/dropin_bench.py [nsteps]"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from cubep3m_amd.kernels import default_tables  # noqa: E402


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    d = os.path.join(ROOT, "oracle", "_ref", "cfg2_8rank")
    exe = os.path.join(d, "hip_mpi_driver")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        raise SystemExit("oracle/_ref/cfg2_8rank/hip_mpi_driver not built (dev container: bash oracle/build_ref.sh)")
    fine, coarse = default_tables()
    os.makedirs(os.path.join(d, "kernels"), exist_ok=True)
    for name, tab in (("wfxyzf.3.ascii", fine), ("wfxyzc.2.ascii", coarse)):
        n = tab.shape[0]
        with open(os.path.join(d, "kernels", name), "w") as f:
            for k in range(n):
                for j in range(n):
                    for i in range(n):
                        f.write("%4d%4d%4d%16.8E%16.8E%16.8E\n" % ((i + 1, j + 1, k + 1) + tuple(float(v) for v in tab[k, j, i])))
    nside, box = 128, 256.0
    scal = np.asarray((0.5, 0.05, 0.05, 8.0), np.float32)
    with tempfile.TemporaryDirectory() as td:
        for r in range(8):
            rng = np.random.default_rng(1000 + r)
            xv = np.zeros((nside ** 3, 6), np.float32)
            xv[:, :3] = np.minimum(rng.random((nside ** 3, 3), dtype=np.float32) * np.float32(box), np.float32(box * (1 - 2e-6)))
            xv[:, 3:] = rng.normal(0, 0.05, (nside ** 3, 3)).astype(np.float32)
            pid = np.arange(1, nside ** 3 + 1, dtype=np.int64) + r * nside ** 3
            with open(os.path.join(td, "in%d.bin" % r), "wb") as f:
                np.asarray([len(xv), nsteps], np.int32).tofile(f)
                scal.tofile(f)
                xv.tofile(f)
                pid.tofile(f)
        res = {}
        for resident in ("1", "0"):
            out = subprocess.run([mpiexec, "-n", "8", exe, td], env=dict(os.environ, OMP_NUM_THREADS="1", P3M_HIP_RESIDENT=resident),
                                 capture_output=True, text=True, timeout=1200)
            if out.returncode != 0:
                raise SystemExit(out.stdout[-3000:] + out.stderr[-3000:])
            ms = [float(l.split(":")[1].split("ms")[0]) for l in out.stdout.splitlines() if "hip_mpi_driver step" in l]
            res[resident] = ms
            n_out = sum(int(np.fromfile(os.path.join(td, "out%d.bin" % r), np.int32, 1)[0]) for r in range(8))
            assert n_out == 8 * nside ** 3, n_out
        for resident, label in (("1", "resident particles (upload at step 1, download at the last step)"), ("0", "copy in / copy out every step")):
            ms = res[resident]
            mid = ms[1:-1] if len(ms) > 2 else ms
            print("%-70s steps: %s  -> middle steps %.1f ms" % (label, " ".join("%.1f" % v for v in ms), float(np.mean(mid))))


if __name__ == "__main__":
    main()
