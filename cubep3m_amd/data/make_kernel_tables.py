"""Convert the reference's Green's-function tables (DATA files, kernels/wfxyzf.3.ascii read at
kernel_initialization.f90:15,25-36 and kernels/wfxyzc.2.ascii read at :344-358) into .npy so the
GPU box (which has no /root/reference) and any host without the ascii files can build kern_f / kern_c.  Run once in the dev container:
    python tests/golden/make_kernel_tables.py
Row format '(3i4,3e16.8)': i j k fx fy fz with i fastest.  Output arrays are [k][j][i][3] float32.
"""
import os

import numpy as np

REF = "/root/reference/kernels"
HERE = os.path.dirname(os.path.abspath(__file__))


def convert(name, n, out):
    t = np.loadtxt(os.path.join(REF, name))
    assert t.shape == (n ** 3, 6)
    i, j, k = (t[:, c].astype(int) - 1 for c in range(3))
    # rows are in loops k (outer), j, i (inner)
    assert np.array_equal(i + n * (j + n * k), np.arange(n ** 3))
    a = t[:, 3:6].astype(np.float32).reshape(n, n, n, 3)
    np.save(os.path.join(HERE, out), a)
    print(out, a.shape, float(abs(a).max()))


if __name__ == "__main__":
    convert("wfxyzf.3.ascii", 16, "wfxyzf3_table.npy")
    convert("wfxyzc.2.ascii", 4, "wfxyzc2_table.npy")
