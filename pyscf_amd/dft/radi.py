"""Radial quadratures and atomic-size adjustments of the Becke partition.

Host-side restatement of ``pyscf/dft/radi.py``: ``gauss_chebyshev`` (:102-117),
``treutler_ahlrichs`` (:139-158, per-element xi table :119-137), ``becke`` (:50-68),
``delley`` (:72-84), ``mura_knowles`` (:87-99), ``becke_atomic_radii_adjust`` (:162-179),
``treutler_atomic_radii_adjust`` (:181-199); Bragg / covalent radii from ``pyscf/data/radii.py:23,53``, SG-1 radii :40.
"""
import numpy as np

from ..gto.mole import BOHR, element_charge as _charge   # radi.py:166,187: radii of the element behind a ghost label

ATOM_SPECIFIC_TREUTLER_GRIDS = True     # radi.py:37

_U = 1.999999                            # data/radii.py:19 (slot 0, never indexed: ghosts use their element)
BRAGG_RADII = 1 / BOHR * np.array((
    _U,
    0.35, 1.40,
    1.45, 1.05, 0.85, 0.70, 0.65, 0.60, 0.50, 1.50,
    1.80, 1.50, 1.25, 1.10, 1.00, 1.00, 1.00, 1.80,
    2.20, 1.80,
    1.60, 1.40, 1.35, 1.40, 1.40, 1.40, 1.35, 1.35, 1.35, 1.35,
    1.30, 1.25, 1.15, 1.15, 1.15, 1.90))

# Cordero et al. covalent radii, H-Kr (pyscf/data/radii.py:53-59)
COVALENT_RADII = 1 / BOHR * np.array((
    _U,
    0.31, 0.28,
    1.28, 0.96, 0.84, 0.73, 0.71, 0.66, 0.57, 0.58,
    1.66, 1.41, 1.21, 1.11, 1.07, 1.05, 1.02, 1.06,
    2.03, 1.76,
    1.70, 1.60, 1.53, 1.39, 1.50, 1.42, 1.38, 1.24, 1.32, 1.22,
    1.22, 1.20, 1.19, 1.20, 1.20, 1.16))

# SG-1 atomic radii in Bohr, H-Ar (radi.py:40-44; Gill, Johnson, Pople, CPL 209, 506)
SG1RADII = np.array((
    1.0000,
    1.0000, 0.5882,
    3.0769, 2.0513, 1.5385, 1.2308, 1.0256, 0.8791, 0.7692, 0.6838,
    4.0909, 3.1579, 2.5714, 2.1687, 1.8750, 1.6514, 1.4754, 1.3333))

_treutler_ahlrichs_xi = [1.0,
    0.8, 0.9,
    1.8, 1.4, 1.3, 1.1, 0.9, 0.9, 0.9, 0.9,
    1.4, 1.3, 1.3, 1.2, 1.1, 1.0, 1.0, 1.0,
    1.5, 1.4,
    1.3, 1.2, 1.2, 1.2, 1.2, 1.2, 1.2, 1.1, 1.1, 1.1,
    1.1, 1.0, 0.9, 0.9, 0.9, 0.9]


def becke(n, charge, *args, **kwargs):
    rm = BRAGG_RADII[charge] if charge == 1 else BRAGG_RADII[charge] * .5
    i = np.arange(n) + 1
    t = np.cos(i * np.pi / (n + 1))
    w = np.pi / (n + 1) * np.sin(i * np.pi / (n + 1))
    r = (1 + t) / (1 - t) * rm
    w = w * 2 / (1 - t) ** 2 * rm
    return r, w


def delley(n, *args, **kwargs):
    r_outer = 12.
    step = 1. / (n + 1)
    i = np.arange(1, n + 1)
    rfac = r_outer / np.log(1 - (n * step) ** 2)
    r = rfac * np.log(1 - (i * step) ** 2)
    dr = rfac * (-2.0 * i * step ** 2) / (1 - (i * step) ** 2)
    return r, dr


def mura_knowles(n, charge=None, *args, **kwargs):
    far = 7 if charge in (3, 4, 11, 12, 19, 20) else 5.2
    x = (np.arange(n) + .5) / n
    r = -far * np.log(1 - x ** 3)
    dr = far * 3 * x * x / ((1 - x ** 3) * n)
    return r, dr


def gauss_chebyshev(n, *args, **kwargs):
    ln2 = 1 / np.log(2)
    fac = 16. / 3 / (n + 1)
    x1 = np.arange(1, n + 1) * np.pi / (n + 1)
    xi = ((n - 1 - np.arange(n) * 2) / (n + 1.) +
          (1 + 2. / 3 * np.sin(x1) ** 2) * np.sin(2 * x1) / np.pi)
    xi = (xi - xi[::-1]) / 2
    r = 1 - np.log(1 + xi) * ln2
    dr = fac * np.sin(x1) ** 4 * ln2 / (1 + xi)
    return r, dr


def treutler_ahlrichs(n, chg, *args, **kwargs):
    xi = _treutler_ahlrichs_xi[chg] if ATOM_SPECIFIC_TREUTLER_GRIDS else 1.
    step = np.pi / (n + 1)
    ln2 = xi / np.log(2)
    i = np.arange(n)
    x = np.cos((i + 1) * step)
    r = -ln2 * (1 + x) ** .6 * np.log((1 - x) / 2)
    dr = step * np.sin((i + 1) * step) * ln2 * (1 + x) ** .6 * (-.6 / (1 + x) * np.log((1 - x) / 2) + 1 / (1 - x))
    return r[::-1], dr[::-1]


treutler = treutler_ahlrichs


def _radii_table(mol, atomic_radii, sqrt):
    charges = [_charge(mol.atom_symbol(i)) for i in range(mol.natm)]
    rad = np.asarray(atomic_radii)[charges]
    rad = (np.sqrt(rad) if sqrt else rad) + 1e-200
    rr = rad.reshape(-1, 1) * (1. / rad)
    a = .25 * (rr.T - rr)
    a[a < -.5] = -.5
    a[a > 0.5] = 0.5
    return a


def becke_atomic_radii_adjust(mol, atomic_radii):
    """-> table a[i,j]; the adjustment is g + a[i,j] (1 - g^2)  (radi.py:162-179)."""
    return _radii_table(mol, atomic_radii, False)


def treutler_atomic_radii_adjust(mol, atomic_radii):
    """radi.py:181-199 (square roots of the radii)."""
    return _radii_table(mol, atomic_radii, True)
