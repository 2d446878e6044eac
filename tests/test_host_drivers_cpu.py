"""Host-side drivers (SCF loop + CDIIS, second-order SCF, gen_response, TDA / TDHF) exercised on the CPU with the oracle's
exact integrals plugged into the same get_jk / one-electron seams the device path fills: the driver logic is then checked
against the reference's known values without a GPU (the device kernels themselves are covered by the -m gpu tests)."""
import numpy as np
import pytest

from oracle import ref


def _oracle_rhf(mol):
    from pyscf_amd import scf

    class OracleRHF(scf.RHF):
        """RHF whose integrals come from the CPU oracle (4-centre ERIs): a test double, not a product path."""

        def __init__(self, mol):
            super().__init__(mol)
            self.with_df = 'oracle'
            self.init_guess = '1e'
            self._eri = ref.int2e(mol)

        def _get_int1e(self):
            if self._int1e is None:
                self._int1e = tuple(ref.int1e(self.mol, k) for k in ('ovlp', 'kin', 'nuc'))
            return self._int1e

        def get_jk(self, mol=None, dm=None, hermi=1, with_j=True, with_k=True, omega=None):
            assert omega is None
            vj, vk = ref.get_jk_exact(self._eri, np.asarray(dm))
            return (vj if with_j else None), (vk if with_k else None)
    return OracleRHF(mol)


@pytest.fixture(scope='module')
def hf_molecule():
    from pyscf_amd import gto
    mol = gto.M(atom=[['H', (0., 0., .917)], ['F', (0., 0., 0.)]], basis='631g')
    mf = _oracle_rhf(mol)
    mf.conv_tol = 1e-12
    mf.kernel()
    assert mf.converged
    return mf


def test_scf_driver_and_newton_reach_the_exact_rhf_golden():
    """H2O / cc-pVDZ exact RHF: -76.026765673119627 (pyscf/scf/test/test_rhf.py) through the product's SCF
    loop (CDIIS, conv_check) and through mf.newton(), both on oracle integrals."""
    from pyscf_amd import gto
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = _oracle_rhf(mol)
    mf.conv_tol = 1e-11
    e = mf.kernel()
    assert mf.converged and abs(e - -76.026765673119627) < 1e-9, e
    mf2 = _oracle_rhf(mol)
    mf2.conv_tol = 1e-11
    nt = mf2.newton()
    e2 = nt.kernel()
    # from the core-Hamiltonian guess the first stationary point is an excited determinant (E = -75.0747): the Aufbau check
    # at the stationary point re-assigns the occupations and the second descent ends on the ground state
    assert nt.converged and abs(e2 - -76.026765673119627) < 1e-9 and nt.cycles <= 20, (e2, nt.cycles)
    assert np.abs(np.sort(mf2.mo_energy) - np.sort(mf.mo_energy)).max() < 1e-6


def test_tda_tdhf_reference_excitation_energies(hf_molecule):
    """pyscf/tdscf/test/test_tdrhf.py:41-74 (HF molecule, 6-31G, exact integrals): TDA / TDHF singlets and triplets to the
    reference's own 5 places in eV."""
    from pyscf_amd import tdscf
    mf = hf_molecule
    for cls, singlet, want in ((tdscf.TDA, True, [11.90276464, 11.90276464, 16.86036434]),
                               (tdscf.TDA, False, [11.01747918, 11.01747918, 13.16955056]),
                               (tdscf.TDHF, True, [11.83487199, 11.83487199, 16.66309285]),
                               (tdscf.TDHF, False, [10.8919234, 10.8919234, 12.63440705])):
        td = cls(mf)
        td.singlet = singlet
        e = td.kernel(nstates=5)[0] * 27.2114
        assert np.abs(e[:3] - want).max() < 2e-5, (cls.__name__, singlet, e[:3])


def test_response_and_ab_matrices_against_explicit_integrals(hf_molecule):
    """A_{ia,jb} = d_ij d_ab (e_a - e_i) + 2 (ia|jb) - (ij|ab), B_{ia,jb} = 2 (ia|bj) - (ib|ja) (triplet: without the
    Coulomb terms) from the MO-transformed oracle ERIs against get_ab() built from gen_response; iterative solvers
    against the dense ones; hermi = 2 keeps only the exchange."""
    from pyscf_amd import tdscf
    mf = hf_molecule
    occ = mf.mo_occ > 0
    co, cv = mf.mo_coeff[:, occ], mf.mo_coeff[:, ~occ]
    eri = mf._eri
    ovov = np.einsum('pqrs,pi,qa,rj,sb->iajb', eri, co, cv, co, cv)
    oovv = np.einsum('pqrs,pi,qj,ra,sb->ijab', eri, co, co, cv, cv)
    de = mf.mo_energy[~occ][None, :] - mf.mo_energy[occ][:, None]
    no, nv = de.shape
    for singlet in (True, False):
        td = tdscf.TDDFT(mf)
        td.singlet = singlet
        a, b = td.get_ab()
        a_ref = -np.einsum('ijab->iajb', oovv)
        b_ref = -np.einsum('ibja->iajb', ovov)
        if singlet:
            a_ref = a_ref + 2 * ovov
            b_ref = b_ref + 2 * ovov
        a_ref = a_ref + np.einsum('ij,ab,ia->iajb', np.eye(no), np.eye(nv), de)
        assert np.abs(a - a_ref).max() < 1e-10 and np.abs(b - b_ref).max() < 1e-10
        e_dense = td.kernel(nstates=3)[0]
        old = tdscf.DENSE_MAX
        tdscf.DENSE_MAX = 0
        try:
            t2 = tdscf.TDDFT(mf)
            t2.singlet = singlet
            e_iter = t2.kernel(nstates=3)[0]
            assert t2.converged.all()
        finally:
            tdscf.DENSE_MAX = old
        assert np.abs(e_iter - e_dense).max() < 1e-6
    x = np.random.default_rng(0).standard_normal((mf.mol.nao, mf.mol.nao))
    anti = (x - x.T) * .5
    v = mf.gen_response(hermi=2)(anti)
    assert np.abs(v + .5 * ref.get_jk_exact(eri, anti)[1]).max() < 1e-12
    v = mf.gen_response(hermi=0)(x)
    vj, vk = ref.get_jk_exact(eri, x)
    assert np.abs(v - (vj - .5 * vk)).max() < 1e-12
    with pytest.raises(TypeError):
        from pyscf_amd.scf._response_functions import gen_uhf_response
        gen_uhf_response(mf)


def test_newton_from_a_reasonable_start_and_restart():
    """The flow of tests/test_gpu_soscf.py on oracle integrals: a stretched water from two DIIS cycles, quadratic
    convergence to the DIIS energy, and a restart from converged orbitals that needs one Fock build and no step."""
    from pyscf_amd import gto
    atoms = [('O', (0., 0., 0.)), ('H', (0., -0.757, 0.587)), ('H', (0.3, 1.1, 0.7))]
    mol = gto.M(atom=atoms, basis='cc-pvdz')
    ref_mf = _oracle_rhf(mol)
    ref_mf.conv_tol = 1e-11
    ref_mf.kernel()
    mf = _oracle_rhf(mol)
    mf.conv_tol = 1e-11
    mf.max_cycle = 2
    mf.kernel()
    assert not mf.converged
    nt = mf.newton()
    e = nt.kernel(mf.mo_coeff, mf.mo_occ)
    assert nt.converged and abs(e - ref_mf.e_tot) < 1e-9 and nt.cycles <= 9 and nt.hessian_products < 80
    assert mf.converged and abs(mf.e_tot - e) < 1e-12
    nt2 = mf.newton()
    e2 = nt2.kernel(mf.mo_coeff, mf.mo_occ)
    assert nt2.cycles == 1 and abs(e2 - e) < 1e-10


def _oracle_uhf(mol):
    from pyscf_amd import scf

    class OracleUHF(scf.UHF):
        """UHF on the oracle's exact integrals (test double)."""

        def __init__(self, mol):
            super().__init__(mol)
            self.with_df = 'oracle'
            self.init_guess = '1e'
            self._eri = ref.int2e(mol)

        def _get_int1e(self):
            if self._int1e is None:
                self._int1e = tuple(ref.int1e(self.mol, k) for k in ('ovlp', 'kin', 'nuc'))
            return self._int1e

        def get_jk(self, mol=None, dm=None, hermi=1, with_j=True, with_k=True, omega=None):
            vj, vk = ref.get_jk_exact(self._eri, np.asarray(dm))
            return (vj if with_j else None), (vk if with_k else None)
    return OracleUHF(mol)


def test_unrestricted_tda_tdhf_reference_excitation_energies(hf_molecule):
    """pyscf/tdscf/test/test_tduhf.py:48-82 (HF molecule, 6-31G): UHF-based TDA / TDHF of the closed-shell molecule (singlets
    and triplets interleaved) and of the spin = 2 state, 4 places in eV; amplitudes normalised to <X|X> - <Y|Y> = 1."""
    from pyscf_amd import gto, tdscf
    mol = gto.M(atom=[['H', (0., 0., .917)], ['F', (0., 0., 0.)]], basis='631g')
    mf = _oracle_uhf(mol)
    mf.conv_tol = 1e-11
    mf.kernel()
    assert mf.converged and abs(mf.e_tot - hf_molecule.e_tot) < 1e-9
    e = tdscf.TDA(mf).kernel(nstates=5)[0] * 27.2114
    assert np.abs(e - [11.01748568, 11.01748568, 11.90277134, 11.90277134, 13.16955369]).max() < 1e-4, e
    td = tdscf.TDHF(mf)
    e, xy = td.kernel(nstates=5)
    assert np.abs(e * 27.2114 - [10.89192986, 10.89192986, 11.83487865, 11.83487865, 12.6344099]).max() < 1e-4, e * 27.2114
    for x, y in xy:
        assert abs((x * x).sum() - (y * y).sum() - 1) < 1e-10
    mol1 = gto.M(atom=[['H', (0., 0., .917)], ['F', (0., 0., 0.)]], basis='631g', spin=2)
    mf1 = _oracle_uhf(mol1)
    mf1.conv_tol = 1e-11
    mf1.kernel()
    assert mf1.converged
    e = tdscf.TDA(mf1).kernel(nstates=5)[0] * 27.2114
    assert np.abs(e - [3.32113736, 18.55977052, 21.01474222, 21.61501962, 25.0938973]).max() < 1e-4, e
    e = tdscf.TDHF(mf1).kernel(nstates=4)[0] * 27.2114
    assert np.abs(e - [3.31267103, 18.4954748, 20.84935404, 21.54808392]).max() < 1e-4, e


def test_cphf_against_finite_field():
    """scf.cphf.solve + gen_vind (pyscf/scf/cphf.py:29-87) on the oracle-integral double: the first-order density of a
    one-electron perturbation equals the finite-field derivative of the SCF density, and 4 sum h1 mo1 the second
    derivative of the energy."""
    from pyscf_amd import gto
    from pyscf_amd.scf import cphf
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='6-31g')
    pert = ref.int1e(mol, 'kin') * 0.3                      # any symmetric one-electron operator

    def scf_at(lam):
        mf = _oracle_rhf(mol)
        mf.conv_tol = 1e-13
        base = tuple(ref.int1e(mol, k) for k in ('ovlp', 'kin', 'nuc'))
        mf._int1e = (base[0], base[1] + lam * pert, base[2])
        mf.kernel()
        assert mf.converged
        return mf
    mf = scf_at(0.0)
    occ = mf.mo_occ > 0
    co, cv = mf.mo_coeff[:, occ], mf.mo_coeff[:, ~occ]
    h1 = cv.T.dot(pert).dot(co)
    mo1, _ = cphf.solve(cphf.gen_vind(mf), mf.mo_energy, mf.mo_occ, h1, tol=1e-11)
    c1 = cv.dot(mo1)
    dm1 = 2 * (c1.dot(co.T) + co.dot(c1.T))
    lam = 1e-4
    mp, mm = scf_at(lam), scf_at(-lam)
    fd = (np.asarray(mp.make_rdm1()) - np.asarray(mm.make_rdm1())) / (2 * lam)
    assert np.abs(fd - dm1).max() < 1e-6, np.abs(fd - dm1).max()
    e2 = (mp.e_tot - 2 * mf.e_tot + mm.e_tot) / lam ** 2
    assert abs(e2 - 4 * (h1 * mo1).sum()) < 2e-4 * abs(e2), (e2, 4 * (h1 * mo1).sum())
    both, _ = cphf.solve(cphf.gen_vind(mf), mf.mo_energy, mf.mo_occ, np.array([h1, -2 * h1]), tol=1e-11)
    assert np.abs(both[0] - mo1).max() < 1e-9 and np.abs(both[1] + 2 * mo1).max() < 1e-9


def test_internal_stability_of_ground_state_and_saddle_point():
    """soscf.stability_rhf_internal: the Aufbau solution of water is a minimum; the stationary point the Newton solver
    reaches from the core-Hamiltonian guess when re-occupation is switched off (E = -75.0747) has a negative Hessian
    eigenvalue."""
    from pyscf_amd import gto, soscf
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = _oracle_rhf(mol)
    mf.conv_tol = 1e-11
    mf.kernel()
    w, stable = soscf.stability_rhf_internal(mf)
    assert stable and w[0] > 0.1
    mf2 = _oracle_rhf(mol)
    mf2.conv_tol = 1e-11
    nt = mf2.newton()
    nt.max_reoccupations = 0
    e = nt.kernel()
    assert nt.converged and abs(e - -75.074736446469) < 1e-8
    w, stable = soscf.stability_rhf_internal(mf2)
    assert not stable and w[0] < -0.1, w


def test_uhf_driver_reference_energies():
    """pyscf/scf/test/test_uhf.py:229-240,478-486 through the product's UHF loop on oracle integrals: the beta-rich triplet
    of water (spin = -2, 6-31G) -75.726396909036637, the helium atom (bare-symbol atom string, STO-3G) -2.8077839575399737,
    and the electron-free He2+ (energy 0)."""
    from pyscf_amd import gto
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='6-31g', spin=-2)
    assert mol.nelec == (4, 6)
    mf = _oracle_uhf(mol)
    mf.conv_tol = 1e-11
    e = mf.kernel()
    assert mf.converged and abs(e - -75.726396909036637) < 1e-9 and mf.mo_occ[1].sum() == 6
    mf = _oracle_uhf(gto.M(atom='He', basis='sto-3g'))
    assert abs(mf.kernel() - -2.8077839575399737) < 1e-12
    mf = _oracle_uhf(gto.M(atom='He', basis='sto-3g', charge=2))
    assert mf.kernel() == 0 and mf.converged
    with pytest.raises(ValueError):
        gto.M(atom='O 0 0', basis='sto-3g')


def _oracle_rohf(mol):
    from pyscf_amd import scf

    class OracleROHF(scf.ROHF):
        """ROHF on the oracle's exact integrals (test double)."""

        def __init__(self, mol):
            super().__init__(mol)
            self.with_df = 'oracle'
            self.init_guess = '1e'
            self._eri = ref.int2e(mol)

        def _get_int1e(self):
            if self._int1e is None:
                self._int1e = tuple(ref.int1e(self.mol, k) for k in ('ovlp', 'kin', 'nuc'))
            return self._int1e

        def get_jk(self, mol=None, dm=None, hermi=1, with_j=True, with_k=True, omega=None):
            vj, vk = ref.get_jk_exact(self._eri, np.asarray(dm))
            return (vj if with_j else None), (vk if with_k else None)
    return OracleROHF(mol)


def test_rohf_driver_reference_energies():
    """pyscf/scf/test/test_h2o_vdz.py:47-61 (H2O+ cc-pVDZ ROHF, exact integrals: -75.627354109594179) and
    test_rhf.py:279-280 (hydrogen atom, STO-3G: -0.46658184955727555) through the product's ROHF loop (Roothaan effective
    Fock matrix, scf/rohf.py) on oracle integrals."""
    from pyscf_amd import gto
    from tests.conftest import H2O
    neutral = _oracle_rhf(gto.M(atom=H2O, basis='cc-pvdz'))
    neutral.kernel()
    dm = np.asarray(neutral.make_rdm1()) * .5       # start from the neutral molecule's density (the reference starts from
    mf = _oracle_rohf(gto.M(atom=H2O, basis='cc-pvdz', charge=1, spin=1))      # minao; the core guess ends on another hole)
    mf.conv_tol = 1e-11
    e = mf.kernel(dm0=np.array((dm, dm)))
    assert mf.converged and abs(e - -75.627354109594179) < 1e-9, e
    mf = _oracle_rohf(gto.M(atom='H', basis='sto-3g', spin=1))
    assert abs(mf.kernel() - -0.46658184955727555) < 1e-12


def test_max_cycle_zero_returns_the_initial_guess_energy():
    """pyscf/scf/hf.py:150-156: max_cycle <= 0 does one eig / get_occ on the initial Fock matrix and returns the
    initial-guess energy (used to crash with UnboundLocalError)."""
    from pyscf_amd import gto
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = _oracle_rhf(mol)
    mf.max_cycle = 0
    e = mf.kernel()
    assert not mf.converged and mf.mo_coeff is not None and mf.mo_occ.sum() == mol.nelectron
    dm0 = mf.get_init_guess(mol, '1e')
    assert abs(e - mf.energy_tot(dm0, mf.get_hcore(), mf.get_veff(mol, dm0))) < 1e-12


def test_scf_reset_drops_cached_integrals():
    """SCF.reset(mol) (pyscf/scf/hf.py:2060-2070): cached one-electron integrals do not survive a new molecule."""
    from pyscf_amd import gto
    from tests.conftest import H2O
    mf = _oracle_rhf(gto.M(atom=H2O, basis='sto-3g'))
    s_a = mf.get_ovlp().copy()
    mol_b = gto.M(atom='O 0 0 0; H 0 -0.9 0.6; H 0 0.9 0.6', basis='sto-3g')
    mf.with_df = None
    mf.reset(mol_b)
    s_b = mf.get_ovlp()
    assert s_a.shape == s_b.shape and np.abs(s_a - s_b).max() > 1e-3


def test_grad_nuc_with_ghost_atom_on_a_nucleus():
    """A ghost atom on top of a real one has zero charge: no 0 * d / 0 in the nuclear repulsion gradient."""
    from pyscf_amd import gto
    from pyscf_amd.grad import rhf as grad_rhf
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587; ghost:H 0 0 0', basis='sto-3g')
    g = grad_rhf.grad_nuc(mol)
    assert np.all(np.isfinite(g)) and np.abs(g[3]).max() == 0
    ref_mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='sto-3g')
    assert np.abs(g[:3] - grad_rhf.grad_nuc(ref_mol)).max() < 1e-14


def test_level_shift_and_damping_reach_the_same_energy():
    """get_fock with damping before DIIS starts and a level shift of the virtual space (pyscf/scf/hf.py:1098-1146,
    :781-805): the converged energy is the exact-RHF golden whatever the convergence aids; the shifted Fock matrix itself
    is F + shift * S (1 - D/2 S)."""
    from pyscf_amd import gto
    from pyscf_amd.scf import hf
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    for kw in (dict(level_shift=0.3), dict(damp=0.5, diis_start_cycle=4), dict(level_shift=0.2, damp=0.3, diis_start_cycle=3)):
        mf = _oracle_rhf(mol)
        mf.conv_tol = 1e-11
        mf.max_cycle = 100
        for k, v in kw.items():
            setattr(mf, k, v)
        e = mf.kernel()
        assert mf.converged and abs(e - -76.026765673119627) < 1e-9, (kw, e)
    s, dm = mf.get_ovlp(), mf.make_rdm1()
    f0 = mf.get_hcore() + mf.get_veff(mol, dm)
    f1 = hf.level_shift(s, dm * .5, f0, 0.7)
    c = mf.mo_coeff
    fmo0, fmo1 = c.T.dot(f0).dot(c), c.T.dot(f1).dot(c)
    nocc = mol.nelectron // 2
    assert np.abs(np.diag(fmo1 - fmo0)[:nocc]).max() < 1e-10 and np.abs(np.diag(fmo1 - fmo0)[nocc:] - 0.7).max() < 1e-10


def test_chkfile_restart(tmp_path):
    """mf.chkfile / init_guess='chkfile' (pyscf/scf/hf.py:679-742, same basis): a restart from the stored orbitals converges
    at once to the same energy."""
    from pyscf_amd import gto
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = _oracle_rhf(mol)
    mf.chkfile = str(tmp_path / 'scf.npz')
    mf.conv_tol = 1e-11
    e = mf.kernel()
    mf2 = _oracle_rhf(mol)
    mf2.chkfile = mf.chkfile
    mf2.init_guess = 'chkfile'
    mf2.conv_tol = 1e-11
    e2 = mf2.kernel()
    assert mf2.converged and abs(e2 - e) < 1e-10 and mf2.cycles <= 2
