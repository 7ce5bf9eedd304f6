"""Generates tests/golden/ref_io.npz from the REFERENCE'S OWN checkpoint.o / particle_initialization.o (oracle/_ref):
the bytes of the checkpoint files it wrote (xv and PID, without and with -DPPINT) and the particles it read from an IC
file written by cubep3m_amd.io_formats.  Dev container only:  python tests/golden/make_ref_io.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)


def main():
    out = {}
    for tag, cfg in (("pm", "cfg1_1rank"), ("pp", "cfg1_pp")):
        with tempfile.TemporaryDirectory() as d:
            f = os.path.join(d, "o.npz")
            subprocess.run([sys.executable, os.path.join(TESTS, "ref_io_run.py"), cfg, f], check=True, stdout=subprocess.DEVNULL,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
            z = np.load(f)
            for k in z.files:
                out["%s_%s" % (tag, k)] = z[k]
    np.savez_compressed(os.path.join(HERE, "ref_io.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
