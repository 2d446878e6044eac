"""CPU oracle for the DF-HF nuclear gradient: central finite differences of the oracle's own DF-RHF / DF-UHF
energies (oracle/ref.py), Richardson-extrapolated (steps h and 2h -> O(h^4)).

TEST INFRASTRUCTURE ONLY (see oracle/ref.py).  What it stands in for: the analytic gradient of
pyscf/df/grad/rhf.py + pyscf/grad/rhf.py (``mf.nuc_grad_method().kernel()``), which needs libcint's derivative
integrals and cannot run here.  Pinned by the reference's own known answers:
pyscf/df/test/test_df_grad.py:61-65 (H2O 6-31G / cc-pvdz-jkfit: lib.fp(g) = 0.005516638190173352 with the
auxiliary-basis response) -- tests/test_oracle_golden.py.
The reference checks its analytic gradient the same way (scanner energies at +-0.001 A, test_df_grad.py:66-75).
"""
import numpy as np

from . import ref


def df_energy(atoms, basis, auxbasis=None, spin=0, charge=0, conv_tol=1e-12, mo0=None):
    """Converged DF-RHF (spin 0) or DF-UHF energy at the geometry atoms = [(symbol, (x, y, z) in Bohr)]."""
    from pyscf_amd import gto
    from pyscf_amd.df import addons
    mol = gto.M(atom=[(s, tuple(r)) for s, r in atoms], basis=basis, unit='Bohr', spin=spin, charge=charge)
    aux = addons.make_auxmol(mol, auxbasis)
    cderi = ref.cholesky_eri(mol, aux)
    if spin == 0:
        def veff(dm, c, occ):
            vj, vk = ref.get_jk(cderi, dm, 1)
            return vj - .5 * vk
        conv, e = ref.rhf_kernel(mol, veff, conv_tol=conv_tol, max_cycle=100)[:2]
    else:
        conv, e = ref.uhf_kernel(mol, cderi, mol.nelec, conv_tol=conv_tol, max_cycle=200, mo0=mo0)
    assert conv
    return e


def fd_gradient(atoms, basis, auxbasis=None, spin=0, charge=0, h=2e-3, components=None, mo0=None):
    """(natm, 3) dE/dR in Eh/Bohr; `components` = iterable of (atom, xyz) to restrict the work (others nan).
    mo0: UHF starting orbitals (Ca, Cb) so that the displaced SCFs stay on the state being differentiated."""
    natm = len(atoms)
    g = np.full((natm, 3), np.nan)
    if components is None:
        components = [(a, x) for a in range(natm) for x in range(3)]

    def e_at(a, x, d):
        moved = [(s, np.array(r, dtype=float)) for s, r in atoms]
        moved[a][1][x] += d
        return df_energy(moved, basis, auxbasis, spin, charge, mo0=mo0)
    for a, x in components:
        d1 = (e_at(a, x, h) - e_at(a, x, -h)) / (2 * h)
        d2 = (e_at(a, x, 2 * h) - e_at(a, x, -2 * h)) / (4 * h)
        g[a, x] = (4 * d1 - d2) / 3
    return g
