"""CPU check of the Rys root/weight Chebyshev tables shipped in libpyscf_amd.so against an
independent evaluation: moments F_k(x) -> exactness of the quadrature for t^(2k), k < 2n."""
import ctypes

import numpy as np
import pytest
from scipy.special import gammainc, gamma

from pyscf_amd import lib as plib

NMAX, DEG, WIDTH = 8, 13, 2.0


def _tables():
    lib = plib.load_library()
    n = lib.PAMD_rys_table_len()
    tab = np.zeros(n)
    off = np.zeros(NMAX + 1, np.int32)
    nint = np.zeros(NMAX + 1, np.int32)
    hu = np.zeros((NMAX + 1, NMAX))
    hw = np.zeros((NMAX + 1, NMAX))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.PAMD_rys_table_host(p(tab), p(off), p(nint), p(hu), p(hw)) == NMAX
    return tab, off, nint, hu, hw


def rys_eval(n, x, T):
    """numpy twin of pamd::rys_root_or_weight (pyscf_amd/csrc/rys_device.h)."""
    tab, off, nint, hu, hw = T
    if x >= nint[n] * WIDTH:
        return hu[n, :n] / x, hw[n, :n] / np.sqrt(x)
    it = min(int(x / WIDTH), nint[n] - 1)
    s = (x - it * WIDTH) * (2.0 / WIDTH) - 1.0
    c = tab[off[n] + it * 2 * n * (DEG + 1): off[n] + (it + 1) * 2 * n * (DEG + 1)].reshape(2 * n, DEG + 1)
    v = np.polynomial.chebyshev.chebval(s, c.T)
    return v[:n], v[n:]


def boys(k, x):
    if x < 1e-12:
        return 1.0 / (2 * k + 1)
    return gammainc(k + .5, x) * gamma(k + .5) / (2 * x ** (k + .5))


@pytest.mark.parametrize('n', range(1, NMAX + 1))
def test_quadrature_reproduces_boys_moments(n):
    T = _tables()
    rng = np.random.default_rng(n)
    xs = np.concatenate([[0.0, 1e-9, 0.5, 1.9999999, 2.0, 2.0000001], rng.uniform(0, 100, 60),
                         [T[2][n] * WIDTH - 1e-9, T[2][n] * WIDTH + 1e-9, 150.0, 1e4]])
    for x in xs:
        u, w = rys_eval(n, x, T)
        assert np.all(u > 0) and np.all(u < 1) and np.all(w > 0)
        for k in range(2 * n):
            ref = boys(k, x)
            got = np.dot(w, u ** k)
            assert abs(got - ref) < 2e-14 * boys(0, x) + 1e-300, (n, x, k, got, ref)
