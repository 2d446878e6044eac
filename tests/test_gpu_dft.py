"""GPU parity of the XC grid path (grids, AO values, nr_rks, DF-RKS energies) vs the CPU oracle
and the reference's golden values."""
import numpy as np
import pytest

from oracle import ref, ref_dft
from tests.conftest import H2O

pytestmark = pytest.mark.gpu

LOWSYM = 'O 0.1 -0.2 0.05; C 0.25 0.4 1.15; H 0.95 -0.3 -0.35'
ATOM_GRID = {'H': (50, 194), 'O': (50, 194)}


def test_golden_grid_norms_and_becke_vs_oracle():
    """pyscf/dft/test/test_grids.py:54-65."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import radi
    mol = gto.M(atom=H2O, basis='6-31g')
    g = dft.Grids(mol)
    g.prune = None
    g.radi_method = radi.gauss_chebyshev
    g.radii_adjust = radi.becke_atomic_radii_adjust
    g.alignment = 0
    g.atom_grid = {'H': (10, 50), 'O': (10, 50)}
    g.build()
    assert abs(np.linalg.norm(g.coords) - 185.91245945279027) < 1e-9
    assert abs(np.linalg.norm(g.weights) - 1720.1317185648893) < 1e-8
    # default scheme (Treutler radial + adjust, NWChem pruning, level 3) vs the numpy partition
    g2 = dft.Grids(mol).build()
    c, w = ref_dft.build_grids(mol)
    assert g2.size == len(w) and g2.size % 8 == 0
    assert np.abs(g2.coords - c).max() < 1e-12
    assert np.abs(g2.weights - w).max() < 1e-11 * np.abs(w).max()


@pytest.mark.parametrize('basis', ['cc-pvtz', '6-31g'])
def test_eval_ao_vs_oracle(basis):
    from pyscf_amd import gto, dft
    mol = gto.M(atom=LOWSYM, basis=basis, spin=1)
    rng = np.random.default_rng(5)
    coords = rng.uniform(-4, 5, (333, 3))
    coords[0] = mol.atom_coords()[0]                  # a point on a nucleus
    ni = dft.NumInt()
    ao = ni.eval_ao(mol, coords, deriv=1)             # (4, ng, nao)
    want = ref_dft.eval_ao(mol, coords, deriv=1)      # (4, ng, nao)
    assert np.abs(ao - want).max() < 1e-12 * max(1, np.abs(want).max())
    ao0 = ni.eval_ao(mol, coords, deriv=0)
    assert np.abs(ao0 - want[0]).max() < 1e-12 * max(1, np.abs(want).max())


@pytest.mark.parametrize('xc', ['lda,vwn', 'lda,vwn_rpa', 'b88,vwn', 'b88,lyp', 'b3lyp', 'pbe,pbe', 'camb3lyp', 'wb97'])
def test_nr_rks_vs_oracle(xc):
    from pyscf_amd import gto, dft, lib
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    grids = dft.Grids(mol)
    grids.atom_grid = (30, 110)
    grids.build()
    hyb, fac = libxc.parse_xc(xc)
    gga = libxc.xc_type(xc) == 'GGA'
    rng = np.random.default_rng(11)
    c = np.linalg.qr(rng.standard_normal((mol.nao, mol.nao)))[0] * 0.7
    occ = np.zeros(mol.nao)
    occ[:5] = 2
    dm = (c * occ).dot(c.T)
    n0, e0, v0 = ref_dft.nr_rks(mol, grids.coords, grids.weights, fac, gga, dm)
    ni = dft.NumInt()
    n1, e1, v1 = ni.nr_rks(mol, grids, xc, dm)                                   # general-DM branch
    n2, e2, v2 = ni.nr_rks(mol, grids, xc, lib.tag_array(dm, mo_coeff=c, mo_occ=occ))   # MO branch
    for n, e, v in ((n1, e1, v1), (n2, e2, v2)):
        assert abs(n - n0) < 1e-10 * abs(n0)
        assert abs(e - e0) < 1e-10 * abs(e0)
        assert np.abs(v - v0).max() < 1e-9 * max(1.0, np.abs(v0).max())
    # several small blocks must give the same result as one block
    ni2 = dft.NumInt(block_bytes=4 * mol.nao * 8 * 1024)
    n3, e3, v3 = ni2.nr_rks(mol, grids, xc, dm)
    assert abs(e3 - e0) < 1e-10 * abs(e0) and np.abs(v3 - v0).max() < 1e-9 * max(1.0, np.abs(v0).max())


def test_golden_df_rks_energy():
    """DF-RKS B88,VWN / 6-31G / 'weigend': -76.690346887915879 (pyscf/dft/test/test_h2o.py:236-240)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import radi, gen_grid
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False            # test_h2o.py:86-89
    try:
        mol = gto.M(atom=H2O, basis='6-31g')
        mf = dft.RKS(mol).density_fit(auxbasis='weigend')
        mf.grids.prune = gen_grid.treutler_prune
        mf.grids.atom_grid = ATOM_GRID
        mf.xc = 'b88, vwn'
        mf.conv_tol = 1e-10
        e = mf.kernel()
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    assert mf.converged and abs(e - -76.690346887915879) < 1e-8, e


def test_df_rks_b3lyp_vs_oracle():
    """Config-3 functional on a small case: DF-RKS B3LYP cc-pVDZ / cc-pvdz-jkfit, level-3 grid."""
    from pyscf_amd import gto, dft, df
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = dft.RKS(mol, xc='b3lyp').density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol, 'cc-pvdz-jkfit'))
    coords, weights = ref_dft.build_grids(mol)
    hyb, fac = libxc.parse_xc('b3lyp')

    def get_jk(dm, c, occ, with_k):
        return ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
    conv, e0 = ref_dft.rks_energy(mol, fac, hyb, True, coords, weights, get_jk)[:2]
    assert conv and abs(e - e0) < 1e-8, (e, e0)


@pytest.mark.parametrize('xc', ['lda,vwn', 'lda,vwn_rpa', 'b88,lyp', 'b3lyp', 'pbe,pbe', 'pbe0', 'camb3lyp', 'wb97'])
def test_nr_uks_vs_oracle(xc):
    """Spin-polarised nr_uks (both branches) vs the oracle whose functionals are pinned by the UKS goldens."""
    from pyscf_amd import gto, dft, lib
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=H2O, basis='cc-pvdz', charge=1, spin=1)
    grids = dft.Grids(mol)
    grids.atom_grid = (30, 110)
    grids.build()
    hyb, fac = libxc.parse_xc(xc)
    gga = libxc.xc_type(xc) == 'GGA'
    rng = np.random.default_rng(3)
    ca = np.linalg.qr(rng.standard_normal((mol.nao, mol.nao)))[0] * 0.7
    cb = np.linalg.qr(rng.standard_normal((mol.nao, mol.nao)))[0] * 0.7
    occ = np.zeros((2, mol.nao))
    occ[0, :5] = 1
    occ[1, :4] = 1
    dms = np.array([(ca * occ[0]).dot(ca.T), (cb * occ[1]).dot(cb.T)])
    n0, e0, v0 = ref_dft.nr_uks(mol, grids.coords, grids.weights, fac, gga, dms[0], dms[1])
    ni = dft.NumInt()
    for d in (dms, lib.tag_array(dms, mo_coeff=np.array([ca, cb]), mo_occ=occ)):
        n1, e1, v1 = ni.nr_uks(mol, grids, xc, d)
        assert np.abs(n1 - np.array(n0)).max() < 1e-10 * max(n0)
        assert abs(e1 - e0) < 1e-10 * abs(e0)
        assert np.abs(v1 - v0).max() < 1e-9 * max(1.0, np.abs(v0).max())
    # closed-shell consistency: nr_uks(D/2, D/2) reproduces nr_rks(D)
    d = dms[0] + dms[1]
    nr, er, vr = ni.nr_rks(mol, grids, xc, d)
    nu, eu, vu = ni.nr_uks(mol, grids, xc, np.array([d * .5, d * .5]))
    assert abs(er - eu) < 1e-10 * abs(er) and np.abs(vu[0] - vr).max() < 1e-9 and np.abs(vu[1] - vr).max() < 1e-9


def test_df_uks_b3lyp_cation_vs_oracle_functional():
    """DF-UKS B3LYP for H2O+ : converged energy vs the oracle energy functional at the same orbitals."""
    from pyscf_amd import gto, dft, df
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=H2O, basis='cc-pvdz', charge=1, spin=1)
    mf = dft.UKS(mol, xc='b3lyp').density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged
    dm = np.asarray(mf.make_rdm1())
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))
    coords, weights = ref_dft.build_grids(mol)
    hyb, fac = libxc.parse_xc('b3lyp')
    n, exc, vxc = ref_dft.nr_uks(mol, coords, weights, fac, True, dm[0], dm[1])
    vj, vk = ref.get_jk(cderi, dm, 1)
    h1e = ref.int1e(mol, 'kin') + ref.int1e(mol, 'nuc')
    vjt = vj[0] + vj[1]
    e0 = (np.einsum('ij,ji', h1e, dm[0] + dm[1]) + .5 * np.einsum('ij,ji', vjt, dm[0] + dm[1]) + exc
          - .5 * hyb * sum(np.einsum('ij,ji', vk[s], dm[s]) for s in range(2)) + mol.energy_nuc())
    assert abs(e - e0) < 1e-8, (e, e0)
    f = h1e + vxc + vjt - hyb * vk
    assert np.linalg.norm(mf.get_grad(mf.mo_coeff, mf.mo_occ, f)) < 1e-4
    assert 0.75 < mf.spin_square() < 0.78


def test_ghost_atom_df_rks_pbe_reference_energy():
    """pyscf/dft/test/test_h2o.py:721-784 (test_ghost_dft_grid): H2O + ghost:H, STO-3G, DF-RKS PBE with
    def2-universal-jkfit, (50,194) grid, three radii-adjust settings; Q-Chem reference -75.2497029684 (2e-5).
    The only in-tree number that pins the PBE exchange-correlation restatement."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import radi
    mol = gto.M(atom="""O 0.000000 0.000000 0.000000
                        H 0.960000 0.000000 0.000000
                        H -0.240000 0.930000 0.000000
                        ghost:H -0.240000 -0.310000 0.880000""", basis='sto-3g')
    assert mol.nelectron == 10 and mol.nao == 8 and mol.atom_charges().tolist() == [8, 1, 1, 0]
    for adjust in (radi.treutler_atomic_radii_adjust, radi.becke_atomic_radii_adjust, None):
        mf = dft.RKS(mol, xc='pbe').density_fit(auxbasis='def2-universal-jkfit')
        mf.grids.atom_grid = (50, 194)
        mf.grids.radii_adjust = adjust
        mf.conv_tol = 1e-10
        e = mf.kernel()
        assert mf.converged and abs(e - -75.2497029684) < 2e-5, (adjust, e)


def test_uks_pbe_closed_shell_limit_equals_rks():
    """Spin-polarised PBE at rho_a = rho_b must reproduce the closed-shell PBE (the one pinned by the reference's
    ghost-atom energy): nr_uks(D/2, D/2) vs nr_rks(D)."""
    from pyscf_amd import gto, dft
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    grids = dft.Grids(mol)
    grids.atom_grid = (30, 110)
    grids.build()
    rng = np.random.default_rng(11)
    c = np.linalg.qr(rng.standard_normal((mol.nao, mol.nao)))[0] * 0.7
    dm = 2 * c[:, :5].dot(c[:, :5].T)
    ni = dft.NumInt()
    n0, e0, v0 = ni.nr_rks(mol, grids, 'pbe,pbe', dm)
    n1, e1, v1 = ni.nr_uks(mol, grids, 'pbe,pbe', (dm * .5, dm * .5))
    assert abs(e1 - e0) < 1e-11 * abs(e0) and abs(n1.sum() - n0) < 1e-10
    assert np.abs(v1[0] - v0).max() < 1e-10 and np.abs(v1[1] - v0).max() < 1e-10


def test_golden_rsh_custom_functional_energy():
    """pyscf/df/test/test_df.py:135-147: HF molecule, cc-pVDZ, DF-RKS with xc = 'lda+0.5*SR_HF(0.3)':
    E = -103.4965622991 (6 places).  Slater exchange plus half of the short-range exact exchange from the
    erfc-attenuated tensor (get_k(omega=-0.3), dft/rks.py:114-117)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import libxc
    assert libxc.parse_xc_rsh('lda+0.5*SR_HF(0.3)')[:3] == (0.5, 0.0, 0.3)
    assert libxc.rsh_coeff('lda+0.5*SR_HF(0.3)') == (0.3, 0.0, 0.5)
    mol = gto.M(atom='H 0 0 0; F 0 0 1.1', basis='ccpvdz')
    mf = dft.RKS(mol, xc='lda+0.5*SR_HF(0.3)').density_fit()
    e = mf.kernel()
    assert mf.converged and abs(e - -103.4965622991) < 5e-7, e
    assert '-0.300000' in mf.with_df._rsh_df                 # the short-range tensor was the one built


def test_golden_eval_gto_fingerprints():
    """pyscf/gto/test/test_eval_gto.py:51-62 on the device: GTOval and GTOval_ip of H2 / cc-pVQZ at 100 seeded points
    (lib.fp = -3.0283379087553808 and -14.526634330008513)."""
    from pyscf_amd import gto, dft
    mol = gto.M(atom='H 0. 0. 0.; H 8. 0. 0.', basis='ccpvqz')
    np.random.seed(1)
    r = np.random.random((100, 3)) * 2
    ao = dft.NumInt().eval_ao(mol, r, deriv=1)
    assert ao.shape == (4, 100, 60)
    assert abs(ref.fp(ao[0]) - -3.0283379087553808) < 1e-10
    assert abs(ref.fp(ao[1:]) - -14.526634330008513) < 1e-9


def _grid_cases():
    from tests.test_oracle_dft_golden import GRID_GOLDENS
    return GRID_GOLDENS


@pytest.mark.parametrize('name,conf,gold', _grid_cases(), ids=[g[0] for g in _grid_cases()])
def test_grid_scheme_goldens_device(name, conf, gold):
    """pyscf/dft/test/test_grids.py:54-115,187-207 through Grids.build (partition on the device, PAMD_grid_partition)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import gen_grid, radi
    from tests.test_oracle_dft_golden import H2O_GRID, check_grid_golden
    mol = gto.M(atom=H2O, basis='6-31g')
    g = dft.Grids(mol)
    g.atom_grid = conf.get('atom_grid', H2O_GRID)
    g.radi_method = getattr(radi, conf.get('radi', 'treutler'))
    g.prune = {None: None, 'sg1': gen_grid.sg1_prune, 'nwchem': gen_grid.nwchem_prune}[conf['prune']]
    adjust = conf.get('adjust', 'treutler')
    g.radii_adjust = None if adjust is None else getattr(radi, adjust + '_atomic_radii_adjust')
    if conf.get('radii') == 'covalent':
        g.atomic_radii = radi.COVALENT_RADII
    g.becke_scheme = {'becke': gen_grid.original_becke, 'stratmann': gen_grid.stratmann}[conf.get('scheme', 'becke')]
    g.alignment = conf.get('alignment', 8)
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = conf.get('specific', False)
    try:
        g.build()
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    check_grid_golden(g.coords, g.weights, gold)


@pytest.mark.parametrize('scheme', ['stratmann', 'lko'])
def test_partition_schemes_vs_oracle(scheme):
    """The Stratmann and Laqua-Kussmann-Ochsenfeld cell functions of PAMD_grid_partition against the numpy restatement
    (gen_grid.py:203-212,388-404; grid_basis.c:236-247,266-384) on a four-element cluster, default grids."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import gen_grid
    mol = gto.M(atom='H 0 0 -0.5; C 0 1 .1; O 0 0 .5; F 1 .3 .5', unit='B', basis='sto-3g')
    g = dft.Grids(mol)
    g.becke_scheme = {'stratmann': gen_grid.stratmann, 'lko': gen_grid.becke_lko}[scheme]
    g.build()
    c, w = ref_dft.build_grids(mol, scheme=scheme)
    assert g.size == len(w)
    assert np.abs(g.coords - c).max() < 1e-12
    assert np.abs(g.weights - w).max() < 1e-11 * np.abs(w).max()
    # quadrature sanity: a normalised s-Gaussian on the carbon atom integrates to 1 with either partition
    r2 = ((g.coords - mol.atom_coords()[1]) ** 2).sum(axis=1)
    assert abs((g.weights * np.exp(-1.3 * r2)).sum() * (1.3 / np.pi) ** 1.5 - 1) < 1e-5


@pytest.mark.parametrize('xc', ['lda,vwn', 'b3lyp', 'pbe,pbe', 'camb3lyp', 'wb97'])
def test_nr_rks_fxc_vs_oracle_and_finite_differences(xc):
    """NumInt.nr_rks_fxc (numint.py:1418-1530): the XC kernel contracted with first-order density matrices, from
    PAMD_eval_fxc (forward-over-forward AD), against (i) the numpy restatement with sympy second derivatives and (ii)
    central differences of the device's own nr_rks potential; hermi = 0 input (only the symmetric part carries a density),
    several matrices per call, HF-only functional = 0."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = dft.RKS(mol, xc='lda,vwn').density_fit().run()
    dm0 = mf.make_rdm1()
    occ = mf.mo_occ > 0
    co, cv = mf.mo_coeff[:, occ], mf.mo_coeff[:, ~occ]
    rng = np.random.default_rng(5)
    a = rng.standard_normal((co.shape[1], co.shape[1]))
    x = rng.standard_normal((co.shape[1], cv.shape[1]))
    d_oo = co.dot(a + a.T).dot(co.T)                      # same decay as rho0: valid for finite differences
    d_ov = co.dot(x).dot(cv.T)                            # non-symmetric occupied-virtual transition density
    grids = dft.Grids(mol)
    grids.atom_grid = (40, 110)
    grids.build()
    ni = dft.NumInt()
    v = ni.nr_rks_fxc(mol, grids, xc, dm0, np.array([d_oo, d_ov, d_ov + d_ov.T]), hermi=0)
    assert v.shape == (3, mol.nao, mol.nao)
    assert np.abs(v - v.transpose(0, 2, 1)).max() < 1e-12
    assert np.abs(v[2] - 2 * v[1]).max() < 1e-10 * np.abs(v[2]).max()
    hyb, fac = libxc.parse_xc(xc)
    gga = libxc.xc_type(xc) == 'GGA'
    n = len(grids.atm_idx)
    for i, d1 in enumerate((d_oo, d_ov)):
        want = ref_dft.nr_rks_fxc(mol, grids.coords[:n], grids.weights[:n], fac, gga, dm0, d1)
        assert np.abs(v[i] - want).max() < 1e-8 * max(1.0, np.abs(want).max()), (xc, i, np.abs(v[i] - want).max())
    eps = 1e-4
    vp = ni.nr_rks(mol, grids, xc, dm0 + eps * d_oo)[2]
    vm = ni.nr_rks(mol, grids, xc, dm0 - eps * d_oo)[2]
    fd = (vp - vm) / (2 * eps)
    assert np.abs(fd - v[0]).max() < 2e-6 * max(1.0, np.abs(v[0]).max()), np.abs(fd - v[0]).max()
    single = ni.nr_rks_fxc(mol, grids, xc, dm0, d_oo, hermi=1)
    assert single.shape == (mol.nao, mol.nao) and np.abs(single - v[0]).max() < 1e-12
    assert np.abs(ni.nr_fxc(mol, grids, 'HF', dm0, d_oo)).max() == 0


@pytest.mark.parametrize('xc', ['lda,vwn', 'b3lyp', 'pbe,pbe', 'camb3lyp', 'wb97'])
def test_nr_uks_fxc_and_singlet_triplet(xc):
    """NumInt.nr_uks_fxc / nr_rks_fxc_st (numint.py:1532-1549,1690-1915) from PAMD_eval_fxc_pol (the spin-polarised
    functionals on nested dual numbers): (i) on a closed-shell reference the singlet kernel equals the closed-shell one,
    nr_rks_fxc(2 dm1) - two independent functional codes; (ii) open-shell cation against the derivative of the oracle's
    nr_uks potential (Richardson finite differences of the sympy restatement); (iii) triplet kernel against the same
    oracle with (dm1, -dm1)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = dft.RKS(mol, xc='lda,vwn').density_fit().run()
    dm0 = mf.make_rdm1()
    occ = mf.mo_occ > 0
    co = mf.mo_coeff[:, occ]
    rng = np.random.default_rng(5)
    a = rng.standard_normal((co.shape[1], co.shape[1]))
    b = rng.standard_normal((co.shape[1], co.shape[1]))
    d1, d2 = co.dot(a + a.T).dot(co.T), co.dot(b + b.T).dot(co.T)
    grids = dft.Grids(mol)
    grids.atom_grid = (40, 110)
    grids.build()
    ni = dft.NumInt()
    hyb, fac = libxc.parse_xc(xc)
    gga = libxc.xc_type(xc) == 'GGA'
    n = len(grids.atm_idx)
    c, w = grids.coords[:n], grids.weights[:n]
    singlet = ni.nr_rks_fxc_st(mol, grids, xc, dm0, np.array([d1, d2]), singlet=True)
    closed = ni.nr_rks_fxc(mol, grids, xc, dm0, 2 * np.array([d1, d2]))
    assert np.abs(singlet - closed).max() < 1e-9 * max(1.0, np.abs(closed).max()), np.abs(singlet - closed).max()
    triplet = ni.nr_rks_fxc_st(mol, grids, xc, dm0, d1, singlet=False)
    want = ref_dft.nr_uks_fxc(mol, c, w, fac, gga, dm0 * .5, dm0 * .5, d1, -d1)[0]
    assert np.abs(triplet - want).max() < 2e-6 * max(1.0, np.abs(want).max()), np.abs(triplet - want).max()
    # open shell: alpha has one more occupied orbital than beta
    da = co.dot(co.T)
    db = co[:, :-1].dot(co[:, :-1].T)
    d1b = co[:, :-1].dot((b + b.T)[:-1, :-1]).dot(co[:, :-1].T)
    v = ni.nr_uks_fxc(mol, grids, xc, (da, db), (d1, d1b))
    assert v.shape == (2, mol.nao, mol.nao)
    want = ref_dft.nr_uks_fxc(mol, c, w, fac, gga, da, db, d1, d1b)
    assert np.abs(v - want).max() < 2e-6 * max(1.0, np.abs(want).max()), np.abs(v - want).max()
    vs = ni.nr_fxc(mol, grids, xc, (da, db), (np.array([d1, d2]), np.array([d1b, d1b])), spin=1)
    assert vs.shape == (2, 2, mol.nao, mol.nao) and np.abs(vs[:, 0] - v).max() < 1e-12


def test_grids_build_with_non0tab_golden():
    """pyscf/dft/test/test_grids.py:132-140 through the PRODUCT's grid builder: `Grids.build()` on the device, then the
    reference's shell mask on the scaled coordinates - 123 non-zero entries, lib.fp = -83.54934301013405; `with_non0tab=True`
    attaches the mask as the reference does (gen_grid.py:615-620)."""
    from pyscf_amd import gto, dft
    from pyscf_amd.dft import gen_grid, radi
    h2o = gto.M(atom=[["O", (0., 0., 0.)], [1, (0., -0.757, 0.587)], [1, (0., 0.757, 0.587)]], basis={"H": '6-31g', "O": '6-31g'})
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    try:
        grid = dft.Grids(h2o)
        grid.atom_grid = {"H": (10, 110), "O": (10, 110)}
        grid.cutoff = 1e-15
        grid.build(with_non0tab=True)
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    non0 = gen_grid.make_mask(h2o, grid.coords * 10.)
    assert (non0 > 0).sum() == 123 and abs(ref.fp(non0) - -83.54934301013405) < 1e-9
    assert grid.non0tab is not None and grid.non0tab.shape == non0.shape and grid.screen_index is grid.non0tab
