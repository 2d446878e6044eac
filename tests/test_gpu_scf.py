"""GPU end-to-end: DF-RHF energies through the reference-style API
(mf = scf.RHF(mol).density_fit(); mf.kernel()) vs golden values and the CPU oracle."""
import numpy as np
import pytest

from oracle import ref

pytestmark = pytest.mark.gpu

H2O = 'O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587'


@pytest.mark.parametrize('basis', ['cc-pvtz', 'cc-pvqz'])
def test_int1e_vs_oracle(basis):
    from pyscf_amd import gto
    from pyscf_amd.scf import hf
    mol = gto.M(atom='O 0.1 -0.2 0.05; C 0.25 0.4 1.15; H 0.95 -0.3 -0.35', basis=basis, spin=1)
    s, t, v = hf.int1e_gpu(mol)
    assert np.abs(s - ref.int1e(mol, 'ovlp')).max() < 1e-12
    assert np.abs(t - ref.int1e(mol, 'kin')).max() < 1e-11
    v0 = ref.int1e(mol, 'nuc')
    assert np.abs(v - v0).max() < 1e-11 * np.abs(v0).max()


def test_golden_df_rhf_energy():
    """E = -76.025936299702536 (pyscf/df/test/test_df_jk.py:57-59), tol 1e-8 Eh."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = scf.RHF(mol).density_fit(auxbasis='weigend')
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged
    assert abs(e - -76.025936299702536) < 1e-8
    assert mf.with_df.get_naoaux() == 71


def test_df_rhf_tz_vs_oracle():
    """cc-pVTZ / cc-pvtz-jkfit (f AOs, g aux): energy within 1e-8 Eh of the oracle SCF."""
    from pyscf_amd import gto, scf, df
    mol = gto.M(atom=H2O, basis='cc-pvtz')
    mf = scf.RHF(mol).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and mf.with_df.auxbasis is None
    aux = df.make_auxmol(mol, 'cc-pvtz-jkfit')
    assert mf.with_df.get_naoaux() == aux.nao == 139
    cderi = ref.cholesky_eri(mol, aux)

    def veff(dm, c, occ):
        vj, vk = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
        return vj - .5 * vk
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
    assert conv and abs(e - e0) < 1e-8, (e, e0)


def test_df_rhf_qz_vs_oracle():
    """cc-pVQZ / cc-pvqz-jkfit (g AOs, h fitting functions; the default pairing of pyscf/df/addons.py:42-72): energy
    within 1e-8 Eh of the oracle SCF, and the tensor itself against the oracle's."""
    from pyscf_amd import gto, scf, df
    mol = gto.M(atom=H2O, basis='cc-pvqz')
    mf = scf.RHF(mol).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    aux = df.make_auxmol(mol, 'cc-pvqz-jkfit')
    assert mf.converged and mol.nao == 115 and mf.with_df.get_naoaux() == aux.nao == 208
    assert mol._bas[:, 1].max() == 4 and aux._bas[:, 1].max() == 5
    cderi = ref.cholesky_eri(mol, aux)
    assert np.abs(mf.with_df._cderi_dev.cpu().numpy() - cderi).max() < 1e-9

    def veff(dm, c, occ):
        vj, vk = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
        return vj - .5 * vk
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
    assert conv and abs(e - e0) < 1e-8, (e, e0)


def test_second_row_rhf_and_b3lyp_vs_oracle():
    """Na-Ar: basis tables, MINAO guess occupations (pyscf/data/elements.py:457-475), Bragg / Treutler radii and the
    period-dependent default grids (pyscf/dft/gen_grid.py:43-60) - H2S, DF-RHF and DF-RKS B3LYP against the oracle
    (own integrals, own grids)."""
    from oracle import ref_dft
    from pyscf_amd import gto, scf, dft, df
    from pyscf_amd.dft import libxc
    mol = gto.M(atom='S 0 0 0.1; H 0 0.96 -0.82; H 0 -0.96 -0.82', basis='cc-pvdz')
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol, 'cc-pvdz-jkfit'))
    mf = scf.RHF(mol).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()

    def veff(dm, c, occ):
        vj, vk = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
        return vj - .5 * vk
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
    assert mf.converged and conv and abs(e - e0) < 1e-8, (e, e0)
    ks = dft.RKS(mol, xc='b3lyp').density_fit()
    ks.conv_tol = 1e-10
    e = ks.kernel()
    coords, weights = ref_dft.build_grids(mol)
    assert ks.grids.size == len(weights) and abs(ks.grids.weights.sum() - weights.sum()) < 1e-7 * weights.sum()
    hyb, fac = libxc.parse_xc('b3lyp')
    conv, e0 = ref_dft.rks_energy(mol, fac, hyb, True, coords, weights,
                                  lambda dm, c, occ, with_k: ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ))[:2]
    assert ks.converged and conv and abs(e - e0) < 1e-8, (e, e0)


def test_fourth_period_rhf_b3lyp_and_gradient_vs_oracle():
    """K-Kr: basis tables, MINAO occupations of the 4th period (pyscf/data/elements.py:582-620), radii and period-dependent
    grids; i fitting shells.  KCl and ZnH2 / def2-SVP: DF-RHF and DF-RKS B3LYP energies against the oracle (own integrals,
    own grids), the ZnH2 DF-RHF gradient against finite differences of the oracle energy."""
    from oracle import ref_dft, ref_grad
    from pyscf_amd import gto, scf, dft, df
    from pyscf_amd.dft import libxc
    B = 0.52917721092
    for atoms in ([('K', (0., 0., 0.)), ('Cl', (0., 0., 2.67))], [('Zn', (0., 0.02, 0.)), ('H', (0., 0.1, 1.55)), ('H', (0.1, 0., -1.5))]):
        mol = gto.M(atom=atoms, basis='def2-svp')
        cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))
        mf = scf.RHF(mol).density_fit()
        mf.conv_tol = 1e-10
        e = mf.kernel()

        def veff(dm, c, occ):
            vj, vk = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
            return vj - .5 * vk
        conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
        assert mf.converged and conv and abs(e - e0) < 1e-8, (atoms[0][0], e, e0)
        if atoms[0][0] == 'Zn':
            g = mf.nuc_grad_method().kernel()
            assert abs(g.sum(axis=0)).max() < 1e-8
            comps = [(0, 2), (1, 1)]
            gfd = ref_grad.fd_gradient([(s, tuple(np.array(x) / B)) for s, x in atoms], 'def2-svp', None, components=comps)
            for a, x in comps:
                assert abs(g[a, x] - gfd[a, x]) < 1e-6, (a, x, g[a, x], gfd[a, x])
        ks = dft.RKS(mol, xc='b3lyp').density_fit()
        ks.conv_tol = 1e-10
        e = ks.kernel()
        coords, weights = ref_dft.build_grids(mol)
        assert ks.grids.size == len(weights)
        hyb, fac = libxc.parse_xc('b3lyp')
        conv, e0 = ref_dft.rks_energy(mol, fac, hyb, True, coords, weights,
                                      lambda dm, c, occ, with_k: ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ))[:2]
        assert ks.converged and conv and abs(e - e0) < 1e-8, (atoms[0][0], e, e0)


def test_golden_minao_guess():
    """Docstring example of init_guess_by_minao (pyscf/scf/hf.py:363-368): H2 / sto-3g."""
    from pyscf_amd import gto
    from pyscf_amd.scf import hf
    mol = gto.M(atom='H 0 0 0; H 0 0 1.1', basis='sto-3g')
    dm = hf.init_guess_by_minao(mol)
    want = np.array([[0.94758917, 0.09227308], [0.09227308, 0.94758917]])
    assert np.abs(dm - want).max() < 1e-8
    # water: the guess carries (about) the right number of electrons
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    dm = hf.init_guess_by_minao(mol)
    s = hf.int1e_gpu(mol)[0]
    assert abs(np.einsum('ij,ji', dm, s) - 10) < 0.05


def test_golden_df_uhf_and_open_shell_vs_oracle():
    """DF-UHF closed shell = -76.025936299702536 (pyscf/df/test/test_df_jk.py:62-64); open-shell cation
    (2 DMs per J/K call through the MO branch) vs the oracle energy functional to 1e-8 Eh."""
    from pyscf_amd import gto, scf, df
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = scf.UHF(mol).density_fit(auxbasis='weigend')
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and abs(e - -76.025936299702536) < 1e-8
    assert abs(mf.spin_square()) < 1e-8
    cat = gto.M(atom=H2O, basis='cc-pvdz', charge=1, spin=1)
    mf = scf.UHF(cat).density_fit(auxbasis='weigend')
    mf.conv_tol = 1e-10
    e = mf.kernel()
    # the oracle's own UHF loop may land on another stationary state for the open-shell cation, so parity
    # is checked at the product's converged orbitals: oracle energy functional and oracle orbital gradient
    cderi = ref.cholesky_eri(cat, df.make_auxmol(cat, 'weigend'))
    dm = np.asarray(mf.make_rdm1())
    h1e = ref.int1e(cat, 'kin') + ref.int1e(cat, 'nuc')
    vj, vk = ref.get_jk(cderi, dm, 1)
    vhf = vj[0] + vj[1] - vk
    e0 = (np.einsum('ij,ji', h1e, dm[0] + dm[1]) + .5 * sum(np.einsum('ij,ji', vhf[s], dm[s]) for s in range(2))
          + cat.energy_nuc())
    assert mf.converged and abs(e - e0) < 1e-8, (e, e0)
    g = mf.get_grad(mf.mo_coeff, mf.mo_occ, h1e + vhf)
    assert np.linalg.norm(g) < 1e-4
    assert 0.75 < mf.spin_square() < 0.77
    assert e < -75.626515724371814          # below the ROHF energy of the same cation (test_df_jk.py:72-78)


def test_golden_df_rohf_cation():
    """DF-ROHF H2O+ / cc-pVDZ / 'weigend': -75.626515724371814 (pyscf/df/test/test_df_jk.py:72-78); the
    K build goes through the ROHF-tagged-DM branch (df_jk.py:346-351)."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='cc-pvdz', charge=1, spin=1)
    mf = scf.ROHF(mol).density_fit(auxbasis='weigend')
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and abs(e - -75.626515724371814) < 1e-8, e
    assert sorted(set(mf.mo_occ.tolist())) == [0.0, 1.0, 2.0]
