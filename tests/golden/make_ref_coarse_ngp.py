"""Generates tests/golden/ref_coarse_ngp.npz from the REFERENCE'S OWN OBJECT CODE built with -DCOARSE_NGP
(oracle/_ref/cfg1_cngp, oracle/build_ref.sh): the coarse deposit (coarse_cic_mass.f90:21-24, coarse_cic_mass_buffer.f90:26-29)
and the coarse kick (coarse_velocity.f90:146-149) with the whole weight on cell i2.  Dev container only:
    python tests/golden/make_ref_coarse_ngp.py
Inputs and the reference's outputs (data only)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))

from common import clustered_particles  # noqa: E402
from make_ref_fixtures import SCAL, TILES, big_stack  # noqa: E402


def main():
    env = dict(os.environ, OMP_NUM_THREADS="1", OMP_STACKSIZE="512M")
    worker = os.path.join(TESTS, "ref_stage_run.py")
    xv = clustered_particles(1500, 64.0, seed=77, frac=0.25, nblobs=6, sigma=1.0, vel_sigma=1.5)
    pid = np.arange(1, 1501, dtype=np.int64) * 7 + 3
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), xv_0=xv, pid_0=pid, scal=SCAL, tiles=TILES)
        subprocess.check_call(["/opt/conda/bin/mpiexec", "-n", "1", sys.executable, worker, "cfg1_cngp", os.path.join(td, "in.npz"), td],
                              env=env, preexec_fn=big_stack, stdout=subprocess.DEVNULL)
        res = dict(np.load(os.path.join(td, "ref_out_0.npz")))
    out = {k: res[k] for k in ("xv_passed", "rho_c", "force_c_halo", "dt_c_acc", "xv_kicked", "xv_final", "pid_final")}
    out.update(xv_in=xv, pid_in=pid, scal=SCAL)
    np.savez_compressed(os.path.join(HERE, "ref_coarse_ngp.npz"), **out)


if __name__ == "__main__":
    main()
