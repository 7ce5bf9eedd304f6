"""The oracle's FFT restates FFTW 2.1.5's rfftwnd semantics (absent third-party library):
unnormalised r2c sign -1 in a (n+2,n,n) in-place array, c2r sign +1, reference divides by n^3
(fftw2.f90:19-22).  Pinned against numpy.fft (layout-identical per SURVEY.md section 8c)."""
import numpy as np
import pytest

import oracle_lib as ol


@pytest.mark.parametrize("n", [8, 12, 20, 28, 40, 44, 80])
def test_r2c_matches_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, n, n)).astype(np.float32)
    a = np.zeros((n, n, n + 2), np.float32)
    a[:, :, :n] = x
    out = ol.fft3d(a, n, +1)
    got = out[:, :, 0::2] + 1j * out[:, :, 1::2]
    ref = np.fft.rfftn(x.astype(np.float64))
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 5e-6, err


@pytest.mark.parametrize("n", [8, 20, 28, 80])
def test_roundtrip_is_identity_with_reference_normalisation(n):
    rng = np.random.default_rng(100 + n)
    x = rng.standard_normal((n, n, n)).astype(np.float32)
    a = np.zeros((n, n, n + 2), np.float32)
    a[:, :, :n] = x
    b = ol.fft3d(ol.fft3d(a, n, +1), n, -1)
    assert np.abs(b[:, :, :n] - x).max() < 2e-5


def test_c2r_matches_numpy():
    n = 20
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, n, n))
    k = np.fft.rfftn(x)
    a = np.zeros((n, n, n + 2), np.float32)
    a[:, :, 0::2] = k.real
    a[:, :, 1::2] = k.imag
    b = ol.fft3d(a, n, -1)
    assert np.abs(b[:, :, :n] - x).max() < 1e-5


def test_delta_function_closed_form():
    n = 12
    a = np.zeros((n, n, n + 2), np.float32)
    a[2, 3, 5] = 1.0  # z=2,y=3,x=5
    out = ol.fft3d(a, n, +1)
    got = out[:, :, 0::2] + 1j * out[:, :, 1::2]
    kz, ky, kx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n // 2 + 1), indexing="ij")
    ref = np.exp(-2j * np.pi * (2 * kz + 3 * ky + 5 * kx) / n)
    assert np.abs(got - ref).max() < 1e-5
