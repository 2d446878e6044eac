"""GPU parity of the in-core 4-centre path (BASELINE config 1: ``scf.RHF(mol)`` without density fitting): the Rys
(ij|kl) kernel and ``dot_eri_dm`` vs the McMurchie-Davidson oracle, and the reference's exact-J/K golden energies
(pyscf/scf/test/test_rhf.py, pyscf/dft/test/test_h2o.py:86-115) through the product."""
import numpy as np
import pytest

from oracle import ref
from tests.conftest import H2O

pytestmark = pytest.mark.gpu

LOWSYM = 'O 0.1 -0.2 0.05; C 0.25 0.4 1.15; H 0.95 -0.3 -0.35'


@pytest.mark.parametrize('atom,basis,spin', [(H2O, 'sto-3g', 0), (H2O, 'cc-pvdz', 0), (LOWSYM, '6-31g**', 1),
                                             (LOWSYM, 'cc-pvtz', 1)])
def test_int2e_vs_oracle(atom, basis, spin):
    """Every (l_i l_j | l_k l_l) class up to (ff|ff), contracted shells, no symmetry in the geometry; all 8 images."""
    from pyscf_amd import gto
    from pyscf_amd.scf import _vhf
    mol = gto.M(atom=atom, basis=basis, spin=spin)
    got = _vhf.int2e_gpu(mol).cpu().numpy()
    want = ref.int2e(mol)
    assert got.shape == want.shape == (mol.nao,) * 4
    assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
    assert np.abs(got - got.transpose(1, 0, 2, 3)).max() == 0 and np.abs(got - got.transpose(2, 3, 0, 1)).max() == 0


def test_range_separated_int2e_and_dot_eri_dm():
    from pyscf_amd import gto
    from pyscf_amd.scf import _vhf
    mol = gto.M(atom=LOWSYM, basis='6-31g', spin=1)
    for omega in (0.4, -0.4):
        got = _vhf.int2e_gpu(mol, omega=omega).cpu().numpy()
        want = ref.int2e(mol, omega)
        assert np.abs(got - want).max() < 1e-11
    eri = _vhf.int2e_gpu(mol)
    rng = np.random.default_rng(1)
    dms = rng.standard_normal((3, mol.nao, mol.nao))                  # non-symmetric, several at once
    vj, vk = _vhf.dot_eri_dm(eri, dms, hermi=0)
    e = ref.int2e(mol)
    assert np.abs(vj - np.einsum('ijkl,xlk->xij', e, dms)).max() < 1e-11
    assert np.abs(vk - np.einsum('ijkl,xjk->xil', e, dms)).max() < 1e-11
    vj1, vk1 = _vhf.dot_eri_dm(eri, dms[0] + 1j * dms[1])
    assert np.abs(vj1 - (vj[0] + 1j * vj[1])).max() < 1e-12 and np.abs(vk1 - (vk[0] + 1j * vk[1])).max() < 1e-12


def test_config1_rhf_goldens():
    """H2O RHF without density fitting: cc-pVDZ -76.026765673119627 (pyscf/scf/test/test_rhf.py), STO-3G
    (BASELINE config 1) against the oracle's exact-ERI SCF; UHF on the closed shell reproduces the RHF energy."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = scf.RHF(mol)
    mf.conv_tol = 1e-11
    e = mf.kernel()
    assert mf.converged and mf.with_df is None and abs(e - -76.026765673119627) < 1e-9, e
    assert abs(scf.UHF(mol).run(conv_tol=1e-11).e_tot - e) < 1e-9
    mol = gto.M(atom=H2O, basis='sto-3g')
    e = scf.RHF(mol).run(conv_tol=1e-11).e_tot
    eri = ref.int2e(mol)

    def veff(dm, c, occ):
        vj, vk = ref.get_jk_exact(eri, dm)
        return vj - .5 * vk
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-11)[:2]
    assert conv and abs(e - e0) < 1e-9, (e, e0)


@pytest.mark.parametrize('xc,e_ref', [('lda,vwn_rpa', -76.01330948329084), ('b88,vwn', -76.690247578608236),
                                      ('b3lypg', -76.384928891413438)])
def test_rks_exact_jk_goldens(xc, e_ref):
    """pyscf/dft/test/test_h2o.py:86-115 (6-31g, (50, 194) grids, Treutler pruning, no atom-specific radial scaling)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import radi
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    try:
        mol = gto.M(atom=H2O, basis='6-31g')
        mf = dft.RKS(mol, xc=xc)
        mf.grids.atom_grid = {'H': (50, 194), 'O': (50, 194)}
        mf.grids.prune = dft.gen_grid.treutler_prune
        mf.conv_tol = 1e-10
        e = mf.kernel()
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    assert mf.converged and abs(e - e_ref) < 2e-8, (xc, e, e_ref)


def test_only_dfj():
    """density_fit(only_dfj=True) (df/df_jk.py:52-54,150-177): fitted J beside the exact 4-centre K."""
    from pyscf_amd import gto, scf, df
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = scf.RHF(mol).density_fit(only_dfj=True)
    mf.conv_tol = 1e-10
    e = mf.kernel()
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))
    eri = ref.int2e(mol)

    def veff(dm, c, occ):
        return ref.get_jk(cderi, dm, 1, with_k=False)[0] - .5 * ref.get_jk_exact(eri, dm)[1]
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
    assert mf.converged and conv and abs(e - e0) < 1e-8, (e, e0)
    e_df = scf.RHF(mol).density_fit().run(conv_tol=1e-10).e_tot
    e_exact = scf.RHF(mol).run(conv_tol=1e-10).e_tot
    assert abs(e - e_df) > 1e-7 and abs(e - e_exact) > 1e-7          # a third, distinct approximation


def test_camb3lyp_goldens():
    """CAM-B3LYP = 0.35 B88 + 0.46 ITYH(omega 0.33) + 0.19 VWN5 + 0.81 LYP with 0.19 short-range / 0.65 long-range exact
    exchange, exact (4-centre) J, K and K_LR: He / cc-pVDZ -2.89299475730048 for RKS and UKS (pyscf/dft/test/test_he.py:87-90),
    H2O / 6-31g (50, 194) grids -76.35549300028714, with omega = 0.15 -76.36649222362115, also through the spelled-out
    functional string (pyscf/dft/test/test_h2o.py:564-577)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import radi
    he = gto.M(atom='He 0 0 0', basis='cc-pvdz')
    assert abs(dft.RKS(he, xc='camb3lyp').run(conv_tol=1e-11).e_tot - -2.89299475730048) < 1e-9
    assert abs(dft.UKS(he, xc='camb3lyp').run(conv_tol=1e-11).e_tot - -2.89299475730048) < 1e-9
    # omega-B97 (attenuated-LSDA B97 exchange, B97 correlation on the original PW92, full long-range exact exchange at 0.4)
    assert abs(dft.RKS(he, xc='wb97').run(conv_tol=1e-11).e_tot - -2.89430888240579) < 1e-9
    assert abs(dft.UKS(he, xc='wb97').run(conv_tol=1e-11).e_tot - -2.89430888240579) < 1e-9
    assert abs(dft.RKS(he, xc='wb97 + 1e-9*HF').run(conv_tol=1e-11).e_tot - -2.89430888240579) < 1e-8
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    try:
        mol = gto.M(atom=H2O, basis='6-31g')
        es = []
        for xc, omega in (('camb3lyp', None), ('camb3lyp', 0.15),
                          ('RSH(.15,0.65,-0.46) + 0.46*ITYH + .35*B88 + VWN5*0.19, LYP*0.81', None)):
            mf = dft.RKS(mol, xc=xc)
            mf.grids.atom_grid = {'H': (50, 194), 'O': (50, 194)}
            mf.conv_tol = 1e-10
            if omega is not None:
                mf.omega = omega
            es.append(mf.kernel())
            assert mf.converged
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    assert abs(es[0] - -76.35549300028714) < 2e-8, es
    assert abs(es[1] - -76.36649222362115) < 2e-8 and abs(es[2] - -76.36649222362115) < 2e-8, es


def test_reference_h2o_631g_scf_goldens():
    """pyscf/scf/test/test_h2o.py:52-107 through the product, exact (in-core 4-centre) and density-fitted ('weigend'): RHF and
    UHF -75.98394849812, ROHF cation -75.578396379589748; DF-RHF / DF-UHF -75.983210886950, DF-ROHF cation
    -75.5775921401438 (all to 1e-9 Eh, conv_tol 1e-11)."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='6-31g')
    cat = gto.M(atom=H2O, basis='6-31g', charge=1, spin=1)
    for make, e_ref in ((lambda: scf.RHF(mol), -75.98394849812), (lambda: scf.UHF(mol), -75.98394849812),
                        (lambda: scf.ROHF(cat), -75.578396379589748),
                        (lambda: scf.RHF(mol).density_fit(auxbasis='weigend'), -75.983210886950),
                        (lambda: scf.UHF(mol).density_fit(auxbasis='weigend'), -75.983210886950),
                        (lambda: scf.ROHF(cat).density_fit(auxbasis='weigend'), -75.5775921401438)):
        mf = make()
        mf.conv_tol = 1e-11
        e = mf.kernel()
        assert mf.converged and abs(e - e_ref) < 1e-9, (type(mf).__name__, mf.with_df is not None, e, e_ref)


@pytest.mark.parametrize('xc,e_ref', [('lda,vwn', -75.350995324984709), ('b3lypg', -75.927304010489976)])
def test_reference_uks_cation_goldens(xc, e_ref):
    """pyscf/dft/test/test_h2o.py:131-142: UKS of H2O+ / 6-31g with exact J/K, (50, 194) grids, Treutler pruning."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import radi
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    try:
        cat = gto.M(atom=H2O, basis='6-31g', charge=1, spin=1)
        mf = dft.UKS(cat, xc=xc)
        mf.grids.atom_grid = {'H': (50, 194), 'O': (50, 194)}
        mf.grids.prune = dft.gen_grid.treutler_prune
        mf.conv_tol = 1e-10
        e = mf.kernel()
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    assert mf.converged and abs(e - e_ref) < 2e-8, (xc, e, e_ref)


def test_integral_direct_jk_equals_incore_and_reference_goldens():
    """The integral-direct branch of RHF.get_jk (pyscf/scf/_vhf.py:370-429 -> CVHFnr_direct_drv, lib/vhf/nr_direct.c:361-489,
    Schwarz prescreen lib/vhf/optimizer.c:90-117): no nao^4 tensor, every shell quartet contracted as it is produced.
      * the reference's exact-integral J/K fingerprints of pyscf/df/test/test_df_jk.py:158-165 (hermi = 0, two random densities),
      * equal to the in-core contraction (and to the oracle's dense tensor) for symmetric and non-symmetric densities,
      * the reference RHF energy of pyscf/scf/test/test_rhf.py:371-372 through an SCF that never builds the tensor,
      * long-range and short-range operators (omega > 0 / < 0)."""
    from pyscf_amd import gto, scf
    from pyscf_amd.scf import _vhf
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    nao = mol.nao
    np.random.seed(1)
    dms = np.random.random((2, nao, nao))
    vj, vk = _vhf.direct(mol, dms, hermi=0)
    assert abs(ref.fp(vj) - -194.08878302990749) < 1e-9 and abs(ref.fp(vk) - -46.530782983591152) < 1e-9
    eri = _vhf.int2e_gpu(mol)
    vj0, vk0 = _vhf.dot_eri_dm(eri, dms, hermi=0)
    assert np.abs(vj - vj0).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11
    sym = dms[0] + dms[0].T
    vj, vk = _vhf.direct(mol, sym, hermi=1)
    vj0, vk0 = _vhf.dot_eri_dm(eri, sym, hermi=1)
    assert np.abs(vj - vj0).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11
    vj1, none = _vhf.direct(mol, sym, with_k=False)
    assert none is None and np.abs(vj1 - vj0).max() < 1e-11
    # screening: a tight cut-off changes nothing beyond the cut-off itself, a crude one only a little
    vj2, vk2 = _vhf.direct(mol, sym, direct_scf_tol=1e-6)
    assert 0 < np.abs(vj2 - vj0).max() + np.abs(vk2 - vk0).max() < 1e-3 or np.abs(vj2 - vj0).max() < 1e-11
    for omega in (0.3, -0.3):
        e_w = _vhf.int2e_gpu(mol, None, omega)
        a = _vhf.direct(mol, dms, hermi=0, omega=omega)
        b = _vhf.dot_eri_dm(e_w, dms, hermi=0)
        assert np.abs(a[0] - b[0]).max() < 1e-11 and np.abs(a[1] - b[1]).max() < 1e-11
    mf = scf.RHF(mol)
    mf.direct_jk = True
    mf.conv_tol = 1e-11
    e = mf.kernel()
    assert mf.converged and not mf._eri                                  # no tensor was built
    assert abs(e - -76.026765673119627) < 1e-9, e
    # d and f shells through the direct kernel
    mol3 = gto.M(atom=H2O, basis='cc-pvtz')
    d3 = np.random.random((mol3.nao, mol3.nao))
    a = _vhf.direct(mol3, d3, hermi=0)
    b = _vhf.dot_eri_dm(_vhf.int2e_gpu(mol3), d3, hermi=0)
    assert np.abs(a[0] - b[0]).max() < 1e-10 and np.abs(a[1] - b[1]).max() < 1e-10
