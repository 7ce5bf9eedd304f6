"""Green's-function tables the host hands to the library.

The reference reads `kernels/wfxyzf.3.ascii` (fine, 16^3 rows; kernel_initialization.f90:15,25-36)
and `kernels/wfxyzc.2.ascii` (coarse, 4^3 rows; :344-358).  A cubep3m host passes its own copies;
`default_tables()` returns the same numbers shipped as .npy data (see data/make_kernel_tables.py).
"""
from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def read_kernel_ascii(path: str, n: int) -> np.ndarray:
    """Parse a '(3i4,3e16.8)' kernel table into float32 [k][j][i][3]."""
    t = np.loadtxt(path)
    if t.shape != (n ** 3, 6):
        raise ValueError(f"{path}: expected {n**3} rows of 6 columns, got {t.shape}")
    i, j, k = (t[:, c].astype(int) - 1 for c in range(3))
    if not np.array_equal(i + n * (j + n * k), np.arange(n ** 3)):
        raise ValueError(f"{path}: error reading in mesh kernel (row order)")
    return np.ascontiguousarray(t[:, 3:6].astype(np.float32).reshape(n, n, n, 3))


def default_tables():
    fine = np.load(os.path.join(_DATA, "wfxyzf3_table.npy"))
    coarse = np.load(os.path.join(_DATA, "wfxyzc2_table.npy"))
    return np.ascontiguousarray(fine, np.float32), np.ascontiguousarray(coarse, np.float32)
