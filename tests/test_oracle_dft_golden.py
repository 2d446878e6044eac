"""Pins the DFT part of the CPU oracle (oracle/ref_dft.py) to the reference's known-answer tests
(pyscf/dft/test/test_h2o.py:86-115,236-240 - note ATOM_SPECIFIC_TREUTLER_GRIDS=False there)."""
import numpy as np
import pytest

from oracle import ref, ref_dft
from tests.conftest import H2O

ATOM_GRID = {'H': (50, 194), 'O': (50, 194)}


@pytest.fixture(scope='module')
def setup():
    from pyscf_amd import gto
    from pyscf_amd.dft import radi
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    mol = gto.M(atom=H2O, basis='6-31g')
    coords, weights = ref_dft.build_grids(mol, ATOM_GRID, prune='treutler')
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    eri = ref.int2e(mol)
    return mol, coords, weights, eri


def _exact_jk(eri):
    def get_jk(dm, c, occ, with_k):
        vj, vk = ref.get_jk_exact(eri, dm)
        return vj, (vk if with_k else None)
    return get_jk


@pytest.mark.parametrize('xc,e_ref', [('lda,vwn_rpa', -76.01330948329084), ('b88,vwn', -76.690247578608236),
                                      ('b3lypg', -76.384928891413438)])
def test_rks_energies_exact_jk(setup, xc, e_ref):
    from pyscf_amd.dft import libxc
    mol, coords, weights, eri = setup
    hyb, fac = libxc.parse_xc(xc)
    gga = libxc.xc_type(xc) == 'GGA'
    conv, e = ref_dft.rks_energy(mol, fac, hyb, gga, coords, weights, _exact_jk(eri))[:2]
    assert conv and abs(e - e_ref) < 2e-8, (xc, e, e_ref)


def test_df_rks_b88vwn(setup):
    """DF-RKS B88,VWN with the 'weigend' fitting basis: -76.690346887915879 (test_h2o.py:236-240)."""
    from pyscf_amd import df
    from pyscf_amd.dft import libxc
    mol, coords, weights, _ = setup
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol, 'weigend'))

    def get_jk(dm, c, occ, with_k):
        return ref.get_jk(cderi, dm, 1, with_k=False)
    hyb, fac = libxc.parse_xc('b88,vwn')
    conv, e = ref_dft.rks_energy(mol, fac, hyb, True, coords, weights, get_jk)[:2]
    assert conv and abs(e - -76.690346887915879) < 2e-8, e


def test_grid_norms():
    """pyscf/dft/test/test_grids.py:54-65: gauss_chebyshev radial, no pruning, Becke radii adjust,
    (10, 50) grids, alignment 0: |coords| = 185.91245945279027, |weights| = 1720.1317185648893."""
    from pyscf_amd import gto
    from pyscf_amd.dft import radi
    mol = gto.M(atom=H2O, basis='6-31g')
    c, w = ref_dft.build_grids(mol, {'H': (10, 50), 'O': (10, 50)}, radi_method=radi.gauss_chebyshev, prune=None,
                               radii_adjust='becke', alignment=0)
    assert abs(np.linalg.norm(c) - 185.91245945279027) < 1e-9
    assert abs(np.linalg.norm(w) - 1720.1317185648893) < 1e-8


@pytest.mark.parametrize('xc,e_ref', [('lda,vwn', -75.350995324984709), ('b3lypg', -75.927304010489976)])
def test_uks_cation_energies_exact_jk(xc, e_ref):
    """Spin-polarised functionals of the oracle vs the reference's UKS goldens for H2O+ / 6-31G
    (pyscf/dft/test/test_h2o.py:131-142): LSDA (Slater + VWN5 with spin stiffness) and B3LYPG
    (polarised B88, LYP and VWN-RPA)."""
    from pyscf_amd import gto
    from pyscf_amd.dft import radi, libxc
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    try:
        cat = gto.M(atom=H2O, basis='6-31g', charge=1, spin=1)
        coords, weights = ref_dft.build_grids(cat, ATOM_GRID, prune='treutler')
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    hyb, fac = libxc.parse_xc(xc)
    conv, e = ref_dft.uks_energy(cat, fac, hyb, libxc.xc_type(xc) == 'GGA', coords, weights, ref.int2e(cat), cat.nelec)
    assert conv and abs(e - e_ref) < 2e-8, (e, e_ref)


def test_oracle_pbe_pinned_by_ghost_atom_reference():
    """pyscf/dft/test/test_h2o.py:721-784: DF-RKS PBE / STO-3G / def2-universal-jkfit on H2O + ghost:H, (50,194) grid;
    reference energy -75.2497029684 (their tolerance 2e-5).  Pins the oracle's PBE exchange and correlation (sympy
    restatement) and the ghost-atom handling of Mole / grids."""
    from pyscf_amd import gto, df
    from pyscf_amd.dft import libxc
    mol = gto.M(atom="""O 0.000000 0.000000 0.000000
                        H 0.960000 0.000000 0.000000
                        H -0.240000 0.930000 0.000000
                        ghost:H -0.240000 -0.310000 0.880000""", basis='sto-3g')
    assert mol.nelectron == 10 and mol.atom_charges().tolist() == [8, 1, 1, 0]
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol, 'def2-universal-jkfit'))
    coords, weights = ref_dft.build_grids(mol, atom_grid=(50, 194))
    hyb, fac = libxc.parse_xc('pbe')

    def get_jk(dm, c, occ, with_k):
        vj, _ = ref.get_jk(cderi, dm, 1, with_k=False)
        return vj, None
    conv, e = ref_dft.rks_energy(mol, fac, hyb, True, coords, weights, get_jk, conv_tol=1e-10)[:2]
    assert conv and abs(e - -75.2497029684) < 2e-5, e


def test_oracle_eval_ao_reference_fingerprints():
    """pyscf/gto/test/test_eval_gto.py:25-32,51-62: H2 (8 Bohr... 8 A apart) cc-pVQZ (s-f shells), 100 seeded points:
    lib.fp(GTOval) = -3.0283379087553808, lib.fp(GTOval_ip) = -14.526634330008513."""
    from pyscf_amd import gto
    mol = gto.M(atom='H 0. 0. 0.; H 8. 0. 0.', basis='ccpvqz')
    assert mol.nao == 60
    np.random.seed(1)
    r = np.random.random((100, 3)) * 2
    ao = ref_dft.eval_ao(mol, r, 1)
    assert abs(ref.fp(ao[0]) - -3.0283379087553808) < 1e-11
    assert abs(ref.fp(ao[1:]) - -14.526634330008513) < 1e-10
