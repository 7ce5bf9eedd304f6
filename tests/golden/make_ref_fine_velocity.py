"""Generates tests/golden/ref_fine_velocity.npz from the REFERENCE'S OWN OBJECT CODE: source_threads/fine_velocity.f90
compiled where it lies into oracle/_ref (oracle/build_ref.sh), builds cfg1_pp (-DNGP -DPPINT), cfg1_1rank (-DNGP) and
cfg1_cic (no -DNGP).  Dev container only:   python tests/golden/make_ref_fine_velocity.py
The fixture holds the input particles and, per build, what fine_velocity left behind: velocities, max |F| per tile,
pp_force_max per tile.  The force box it ran on is synthetic and regenerated from its seed (tests/ref_fv_run.py)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))

from test_oracle_vs_ref import CHILD_ENV, SCAL, _big_stack, fine_velocity_input  # noqa: E402


def main():
    xv, pid = fine_velocity_input(n=1500, seed=41)
    out = {"xv_in": xv, "pid_in": pid, "scal": np.asarray(SCAL, np.float32)}
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), xv=xv, pid=pid, scal=out["scal"])
        for cfg in ("cfg1_pp", "cfg1_1rank", "cfg1_cic"):
            subprocess.check_call([sys.executable, os.path.join(TESTS, "ref_fv_run.py"), cfg, os.path.join(td, "in.npz"), os.path.join(td, "o.npz")],
                                  stdout=subprocess.DEVNULL, preexec_fn=_big_stack, env=CHILD_ENV)
            r = np.load(os.path.join(td, "o.npz"))
            # ghosts are never kicked by fine_velocity: keep the first np_local records only
            out[cfg + "_vel"] = r["xv_kicked"][: len(xv), 3:]
            assert np.array_equal(r["xv_kicked"][: len(xv), :3], xv[:, :3]) and np.array_equal(r["pid_kicked"][: len(xv)], pid)
            assert np.array_equal(r["xv_kicked"][len(xv):, 3:], r["xv_kicked"][len(xv):, 3:])
            out[cfg + "_np_passed"] = np.int32(len(r["xv_kicked"]))
            out[cfg + "_f_force_max"] = r["f_force_max"]
            out[cfg + "_pp_force_max"] = r["pp_force_max"]
    np.savez_compressed(os.path.join(HERE, "ref_fine_velocity.npz"), **out)
    print("ref_fine_velocity.npz", os.path.getsize(os.path.join(HERE, "ref_fine_velocity.npz")))


if __name__ == "__main__":
    main()
