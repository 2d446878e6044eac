"""GPU parity: Rys-quadrature int3c2e / int2c2e kernels and the cderi build vs the CPU oracle
(McMurchie-Davidson, independently pinned to the reference's golden fingerprints)."""
import numpy as np
import pytest

from oracle import ref

pytestmark = pytest.mark.gpu

LOWSYM = 'O 0.1 -0.2 0.05; C 0.25 0.4 1.15; H 0.95 -0.3 -0.35'


def _dev():
    import torch
    return torch.device('cuda', 0)


def test_golden_int3c2e_fingerprint(h2o_dz):
    """lib.fp(int3c2e_sph, s2ij) = 12.407403711205063 (pyscf/df/test/test_incore.py:72)."""
    from pyscf_amd.df import incore
    mol, aux = h2o_dz
    j3c = incore.aux_e2_gpu(mol, aux, _dev()).cpu().numpy()       # (naux, npair)
    assert abs(ref.fp(j3c.T) - 12.407403711205063) < 1e-9
    want = ref.pack_tril(ref.int3c2e(mol, aux))
    assert np.abs(j3c - want).max() < 1e-12


def test_int2c2e_and_cderi(h2o_dz):
    from pyscf_amd.df import incore
    from pyscf_amd.gto.moleintor import IntEngine
    mol, aux = h2o_dz
    eng = IntEngine(mol, aux, _dev())
    j2c = eng.int2c2e().cpu().numpy()
    want = ref.int2c2e(aux)
    assert np.abs(j2c - want).max() < 1e-12 * np.abs(want).max()
    cderi = incore.cholesky_eri_gpu(mol, aux, _dev()).cpu().numpy()
    want = ref.cholesky_eri(mol, aux)
    assert np.abs(cderi - want).max() < 1e-10
    # sharded build: rows [l0,l1) of the same tensor
    part = incore.cholesky_eri_gpu(mol, aux, _dev(), 20, 55).cpu().numpy()
    assert np.abs(part - want[20:55]).max() < 1e-10
    # several AO-row slabs must give the same tensor
    small = incore.cholesky_eri_gpu(mol, aux, _dev(), slab_bytes=40 * aux.nao * 8).cpu().numpy()
    assert np.abs(small - want).max() < 1e-10


@pytest.mark.parametrize('basis,auxbasis', [('cc-pvtz', 'cc-pvtz-jkfit'), ('def2-tzvp', 'def2-universal-jkfit'),
                                            ('sto-3g', 'weigend'), ('cc-pvqz', 'cc-pvqz-jkfit'),
                                            ('def2-qzvpp', 'def2-universal-jkfit')])
def test_all_classes_low_symmetry(basis, auxbasis):
    """AO l<=3 (f) and aux l<=4 (g) for the triple-zeta sets, AO l<=4 (g) and aux l<=5 (h) for the quadruple-zeta ones
    (7 Rys roots for (gg|h)); contracted aux primitives, no symmetry in the geometry: exercises every (l_i, l_j | l_k)
    kernel of the family."""
    from pyscf_amd import gto
    from pyscf_amd.df import incore
    mol = gto.M(atom=LOWSYM, basis=basis, spin=1)
    aux = gto.M(atom=LOWSYM, basis=auxbasis, spin=1)
    got = incore.aux_e2_gpu(mol, aux, _dev()).cpu().numpy()
    want = ref.pack_tril(ref.int3c2e(mol, aux))
    err = np.abs(got - want).max()
    assert err < 1e-11 * max(1.0, np.abs(want).max()), err


def test_fourth_period_classes_with_i_fitting_shells():
    """3d metal with the def2 sets: AO shells up to f, fitting shells up to i (l = 6; def2-universal-jkfit of Sc-Zn), no symmetry:
    the (l_i l_j | 6) kernels, the (6, 0 | l) metric classes, and the tensor they build."""
    from pyscf_amd import gto, df
    from pyscf_amd.df import incore
    atom = 'Zn 0.1 -0.2 0.05; O 0.25 0.4 1.85; H 0.95 -0.3 2.35'
    mol = gto.M(atom=atom, basis='def2-tzvp', spin=1)
    aux = df.make_auxmol(mol)
    assert mol._bas[:, 1].max() == 3 and aux._bas[:, 1].max() == 6
    got = incore.aux_e2_gpu(mol, aux, _dev()).cpu().numpy()
    want = ref.pack_tril(ref.int3c2e(mol, aux))
    assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
    cderi = incore.cholesky_eri_gpu(mol, aux, _dev()).cpu().numpy()
    assert np.abs(cderi - ref.cholesky_eri(mol, aux)).max() < 1e-9


def test_long_range_integrals_and_rsh_get_jk(h2o_dz):
    """omega > 0: erf(omega r12)/r12 integrals vs the oracle (tight), and DF.get_jk(dm, omega=1.1) vs the
    reference fingerprints (3 places, pyscf/df/test/test_df.py:101-117)."""
    from pyscf_amd import gto, df
    from pyscf_amd.df import incore
    from pyscf_amd.gto.moleintor import IntEngine
    from tests.conftest import H2O
    mol = gto.M(atom=LOWSYM, basis='cc-pvtz', spin=1)
    aux = gto.M(atom=LOWSYM, basis='cc-pvtz-jkfit', spin=1)
    for omega in (0.3, 1.1):
        got = incore.aux_e2_gpu(mol, aux, _dev(), omega=omega).cpu().numpy()
        want = ref.pack_tril(ref.int3c2e(mol, aux, omega))
        assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
        j2c = IntEngine(mol, aux, _dev(), omega).int2c2e().cpu().numpy()
        w2 = ref.int2c2e(aux, omega)
        assert np.abs(j2c - w2).max() < 1e-11 * np.abs(w2).max()
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    obj = df.DF(mol)
    np.random.seed(1)
    dm = np.random.random((2, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dm, hermi=0, omega=1.1)
    assert abs(ref.fp(vj) - -181.5033531437091) < 5e-4
    assert abs(ref.fp(vk) - -37.78854217974532) < 5e-4
    vj0, vk0 = obj.get_jk(dm, hermi=0)                       # the Coulomb tensor is a separate object
    assert abs(ref.fp(vj0) - ref.fp(vj)) > 1.0


def test_short_range_integrals_and_get_jk():
    """omega < 0 (libcint's convention for erfc(|omega| r12)/r12, pyscf/gto/mole.py:76-84 - what get_veff asks for when
    a functional has short-range exact exchange only, dft/rks.py:114-117): 3- and 2-centre integrals vs the oracle,
    SR + LR = Coulomb at the integral level, DF.get_jk(omega < 0) and the integral-direct J vs the oracle's short-range
    Cholesky tensor."""
    from pyscf_amd import gto, df
    from pyscf_amd.df import incore, df_jk
    from pyscf_amd.gto.moleintor import IntEngine
    from tests.conftest import H2O
    mol = gto.M(atom=LOWSYM, basis='cc-pvtz', spin=1)
    aux = gto.M(atom=LOWSYM, basis='cc-pvtz-jkfit', spin=1)
    got = incore.aux_e2_gpu(mol, aux, _dev(), omega=-0.3).cpu().numpy()
    want = ref.pack_tril(ref.int3c2e(mol, aux, -0.3))
    assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())
    full = incore.aux_e2_gpu(mol, aux, _dev()).cpu().numpy()
    lr = incore.aux_e2_gpu(mol, aux, _dev(), omega=0.3).cpu().numpy()
    assert np.abs(got + lr - full).max() < 1e-12 * np.abs(full).max()
    j2c = IntEngine(mol, aux, _dev(), -0.3).int2c2e().cpu().numpy()
    w2 = ref.int2c2e(aux, -0.3)
    assert np.abs(j2c - w2).max() < 1e-11 * np.abs(w2).max()
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    obj = df.DF(mol)
    np.random.seed(1)
    dm = np.random.random((2, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dm, hermi=0, omega=-0.5)
    cd = ref.cholesky_eri(mol, df.make_auxmol(mol), omega=-0.5)
    vj1, vk1 = ref.get_jk(cd, dm, hermi=0)
    assert np.abs(vj - vj1).max() < 1e-9 and np.abs(vk - vk1).max() < 1e-9
    dsym = dm[0] + dm[0].T
    vjd = df_jk.get_j(obj.range_coulomb(-0.5), dsym)          # integral-direct J generates the same short-range slabs
    assert np.abs(vjd - ref.get_jk(cd, dsym, hermi=1, with_k=False)[0]).max() < 1e-9


def test_benzene_overlap_and_2c2e_reference_fingerprints_device():
    """pyscf/gto/test/test_moleintor.py:94-96,331-333 on the device kernels: sum |S| = 622.29059965181796 and
    lib.fp(int2c2e over the AO shells of labelled benzene / cc-pVDZ) = -460.83033192375615."""
    from pyscf_amd import gto
    from pyscf_amd.gto.moleintor import IntEngine
    from pyscf_amd.scf.hf import int1e_gpu
    from tests.test_oracle_golden import BENZENE_LABELLED, BENZENE_BASIS
    mol = gto.M(atom=BENZENE_LABELLED, basis=BENZENE_BASIS)
    s = int1e_gpu(mol, _dev())[0]
    assert abs(np.abs(s).sum() - 622.29059965181796) < 1e-9
    j2c = IntEngine(mol, mol, _dev()).int2c2e().cpu().numpy()
    assert abs(ref.fp(j2c) - -460.83033192375615) < 1e-8
