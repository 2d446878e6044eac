"""oracle/_ref - the reference's own C for the DF J/K hot path (pyscf/lib/ao2mo/nr_ao2mo.c:399-419,1016-1031,1240-1266 and
pyscf/lib/np_helper/{pack_tril,npdot}.c, compiled in place by `make -C oracle ref`) - against the numpy restatement in
oracle/ref.py and against the reference's golden fingerprints.  This is what bench.py's cpu_baseline (kind "reference") times."""
import numpy as np
import pytest

from oracle import ref, ref_c
from pyscf_amd import gto, df
from tests.conftest import H2O

pytestmark = pytest.mark.skipif(not ref_c.build(), reason='oracle/_ref not built (needs /root/reference or the prebuilt .so)')


def _case(basis='cc-pvdz', aux='weigend', seed=3):
    mol = gto.M(atom=H2O, basis=basis)
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol, aux))
    nao, nocc = mol.nao, mol.nelectron // 2
    c = np.linalg.qr(np.random.RandomState(seed).rand(nao, nao))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    return mol, cderi, c, occ, (c[:, :nocc] * 2).dot(c[:, :nocc].T)


def test_reference_c_jk_equals_restatement():
    mol, cderi, c, occ, dm = _case()
    vj0, vk0 = ref.get_jk(cderi, dm, hermi=1)                         # general-DM branch of the restatement
    for blk in (240, 50, 7):                                         # several blocks, ragged last block
        vj, vk, flops = ref_c.get_jk(cderi, dm, c, occ, blockdim=blk, nthreads=2)
        assert np.abs(vj - vj0).max() < 1e-12 and np.abs(vk - vk0).max() < 1e-12
    assert flops > 0 and set(ref_c.get_jk.last_phases) == {'pack', 'vj', 'e2_drv', 'dot', 'unpack'}


def test_reference_c_pack_unpack_and_npdgemm():
    rng = np.random.RandomState(0)
    a = rng.rand(3, 9, 9)
    a = a + a.transpose(0, 2, 1)
    t = ref_c.pack_tril(a)
    assert np.array_equal(t, ref.pack_tril(a)) and np.array_equal(ref_c.unpack_tril(t), a)
    for k, n in ((400, 5), (7, 11)):                                  # both NPdgemm strategies (parallel over k / plain)
        b = rng.rand(k, n)
        assert np.abs(ref_c.lib_dot_tn(b) - b.T.dot(b)).max() < 1e-12


def test_reference_c_reproduces_golden_energy_ingredients():
    """E(DF-RHF) of the reference's test_df_jk.py:57-59 is built from these J/K: the _ref path must give the restatement's
    converged energy ingredients at the converged density."""
    mol, cderi, c, occ, dm = _case(aux='weigend')
    vj, vk, _ = ref_c.get_jk(cderi, dm, c, occ)
    vj0, vk0 = ref.get_jk(cderi, dm, hermi=1)
    assert abs(np.einsum('ij,ji', dm, vj - .5 * vk) - np.einsum('ij,ji', dm, vj0 - .5 * vk0)) < 1e-11
