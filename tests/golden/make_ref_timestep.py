"""Generates tests/golden/ref_timestep.npz from the REFERENCE'S OWN timestep.f90 object code (oracle/_ref, built by
oracle/build_ref.sh from /root/reference/source_threads where it lies).  Dev container only:
    python tests/golden/make_ref_timestep.py
Data only: per-step (a, a_mid, da, dt, dt_old, dt_gas, tau, t | nts, checkpoint/projection/halofind/final flags) of the
scenarios in tests/time_scenarios.py and a table of expansion(a0, dt0) -> (da1, da2)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)


def main():
    out = {}
    # label, reference build, scenario of tests/time_scenarios.py ("chap": the -DChaplygin build over the PM-only scenario)
    for name, cfg, scen in (("pm", "cfg1_1rank", "pm"), ("pp", "cfg1_pp", "pp"), ("chap", "cfg1_chap", "pm")):
        with tempfile.TemporaryDirectory() as d:
            f = os.path.join(d, "o.npz")
            subprocess.run([sys.executable, os.path.join(TESTS, "ref_time_run.py"), cfg, scen, f], check=True, stdout=subprocess.DEVNULL,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
            z = np.load(f)
            for k in z.files:
                out["%s_%s" % (name, k)] = z[k]
    np.savez_compressed(os.path.join(HERE, "ref_timestep.npz"), **out)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
