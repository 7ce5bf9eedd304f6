"""Generates tests/golden/ref_projection.npz from the REFERENCE'S OWN projection.o (oracle/_ref): the three density
projections of a seeded clustered particle set and the bytes of the files it wrote.
Dev container only:  python tests/golden/make_ref_projection.py"""
import os
import resource
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)


def big_stack():   # the reference keeps large automatic arrays on the stack (particle_pass)
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


def main():
    out = {}
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "o.npz")
        subprocess.run([sys.executable, os.path.join(TESTS, "ref_proj_run.py"), "cfg1_1rank", f], check=True, stdout=subprocess.DEVNULL,
                       env=dict(os.environ, OMP_NUM_THREADS="1"), preexec_fn=big_stack)
        z = np.load(f)
        for k in z.files:
            out[k] = z[k]
    np.savez_compressed(os.path.join(HERE, "ref_projection.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
