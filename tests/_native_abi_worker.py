"""Worker of tests/test_gpu_native_abi.py: drives the host-array C ABI (PAMD_df_create / PAMD_df_get_jk / PAMD_df_export_cderi)
from plain numpy + ctypes.  torch is never imported in this process (asserted), the caller owns every buffer - the reference's
convention (pyscf/df/df_jk.py:373-379).  Checks: the reference's own golden fingerprints (pyscf/df/test/test_df_jk.py:144-156,
pyscf/df/test/test_df.py:53), the oracle's tensor / J / K on the MO branch (fused and unfused J), a linearly dependent metric."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import time as _t
    _t0 = _t.perf_counter()

    def stamp(what):
        print('[%7.2f s] %s' % (_t.perf_counter() - _t0, what), flush=True)
    from oracle import ref
    from pyscf_amd import gto, lib
    from pyscf_amd.df import addons, native
    assert 'torch' not in sys.modules
    stamp('imports done')
    h2o = 'O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587'
    mol = gto.M(atom=h2o, basis='cc-pvdz')
    import time
    t0 = time.perf_counter()
    obj = native.NativeDF(mol, auxbasis='weigend').build()
    print('first PAMD_df_create (HIP + rocBLAS / rocSOLVER start-up included): %.2f s' % (time.perf_counter() - t0), flush=True)
    nao = mol.nao
    assert obj.get_naoaux() == 71
    stamp('first handle built')
    assert native.NativeDF(mol).get_naoaux() == 116                        # default aux basis, test_df.py:53
    # G4: general-DM branch, two densities
    np.random.seed(1)
    dms = np.random.random((2, nao, nao))
    vj, vk = obj.get_jk(dms, hermi=0)
    assert abs(lib.fp(vj) - -194.15910890730066) < 1e-9, lib.fp(vj)
    assert abs(lib.fp(vk) - -46.365071587653517) < 1e-9, lib.fp(vk)
    vj1, none = obj.get_jk(dms, hermi=0, with_k=False)
    none2, vk1 = obj.get_jk(dms, hermi=0, with_j=False)
    assert none is None and none2 is None and np.abs(vj1 - vj).max() < 1e-12 and np.abs(vk1 - vk).max() < 1e-12
    stamp('G4 / G6 goldens done')
    # the tensor itself, row block by row block (DF.loop)
    aux = addons.make_auxmol(mol, 'weigend')
    cderi = ref.cholesky_eri(mol, aux)
    got = np.vstack(list(obj.loop(20)))
    assert got.shape == cderi.shape and np.abs(got - cderi).max() < 1e-10
    # MO branch (tagged density), J fused into the half transform (promise) and unfused (tag that does not match)
    for basis, nw in (('cc-pvdz', 1), ('cc-pvtz', 2)):
        from pyscf_amd.data import clusters
        m2 = gto.M(atom=clusters.water_cluster(nw), basis=basis)
        t0 = time.perf_counter()
        o2 = native.NativeDF(m2).build()
        print('PAMD_df_create %s (H2O)_%d: %.2f s' % (basis, nw, time.perf_counter() - t0), flush=True)
        cd = ref.cholesky_eri(m2, addons.make_auxmol(m2))
        n, nocc = m2.nao, m2.nelectron // 2
        c = np.linalg.qr(np.random.RandomState(3).rand(n, n))[0]
        occ = np.zeros(n)
        occ[:nocc] = 2
        dm = (c[:, :nocc] * 2).dot(c[:, :nocc].T)
        vj0, vk0 = ref.get_jk(cd, dm, 1, mo_coeff=c, mo_occ=occ)
        vj, vk = o2.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
        assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9, (basis, np.abs(vj - vj0).max(), np.abs(vk - vk0).max())
        dm_other = dm + 0.01 * np.eye(n)                                   # tag inconsistent with the matrix: J must follow dm
        vj, vk = o2.get_jk(lib.tag_array(dm_other, mo_coeff=c, mo_occ=occ), hermi=1)
        vj0b, _ = ref.get_jk(cd, dm_other, 1)
        assert np.abs(vj - vj0b).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9
        assert o2._last_mismatch > 1e-6 and not o2._last_fused          # r06: found by the probe INSIDE the call (flags bit 1)
        # the advisor's finite-difference edit of a tagged array: two elements, caught by the full-matrix probe
        fd = dm.copy()
        fd[1, 2] += 1e-4
        fd[2, 1] += 1e-4
        vj, vk = o2.get_jk(lib.tag_array(fd, mo_coeff=c, mo_occ=occ, dm_from_orbitals=True), hermi=1)
        vj0c, _ = ref.get_jk(cd, fd, 1)
        assert np.abs(vj - vj0c).max() < 1e-9 and o2._last_mismatch > 1e-8
        vj, vk = o2.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
        assert o2._last_fused and o2._last_mismatch < 1e-12 and np.abs(vj - vj0).max() < 1e-9
        # UHF-style: two densities with their own orbitals
        occ_b = np.zeros(n)
        occ_b[:nocc - 1] = 1
        occ_a = np.zeros(n)
        occ_a[:nocc] = 1
        dms2 = np.array([(c[:, :nocc]).dot(c[:, :nocc].T), (c[:, :nocc - 1]).dot(c[:, :nocc - 1].T)])
        vj, vk = o2.get_jk(lib.tag_array(dms2, mo_coeff=np.array([c, c]), mo_occ=np.array([occ_a, occ_b])), hermi=1)
        vj0, vk0 = ref.get_jk(cd, dms2, 1)
        assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9
        o2.reset()
    stamp('MO branch cases done')
    # no square image (PAMD_DF_SQUARE=0 = DF.k_square False): packed-operand half transform with the diagonal-block side image
    # (PAMD_nr_e2_symm_diag) in both K branches - 120 fractionally occupied orbitals so that the 128-orbital chunk shape applies
    os.environ['PAMD_DF_SQUARE'] = '0'
    try:
        m4 = gto.M(atom=clusters.water_cluster(4), basis='cc-pvtz')
        o4 = native.NativeDF(m4).build()
        cd = ref.cholesky_eri(m4, addons.make_auxmol(m4))
        n = m4.nao
        assert n >= 197
        c = np.linalg.qr(np.random.RandomState(5).rand(n, n))[0]
        occ = np.zeros(n)
        occ[:120] = np.linspace(2.0, 0.5, 120)
        dm = (c * occ).dot(c.T)
        vj0, vk0 = ref.get_jk(cd, dm, 1, mo_coeff=c, mo_occ=occ)
        vj, vk = o4.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
        assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9, (np.abs(vj - vj0).max(), np.abs(vk - vk0).max())
        for sched in ('serial', 'overlap'):                                   # both schedules of the fused path's second J pass
            os.environ['PAMD_DF_J2'] = sched
            vj, vk = o4.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
            assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9, sched
        del os.environ['PAMD_DF_J2']
        dmg = dm + 0.05 * np.random.RandomState(6).rand(n, n)                 # no tag, not symmetric: general branch
        vj0, vk0 = ref.get_jk(cd, dmg, 0)
        vj, vk = o4.get_jk(dmg, hermi=0)
        assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9, (np.abs(vj - vj0).max(), np.abs(vk - vk0).max())
        o4.reset()
    finally:
        del os.environ['PAMD_DF_SQUARE']
    stamp('packed + diagonal-block image cases done')
    # linearly dependent metric: two copies of the aux basis on the same atoms -> Cholesky fails, eigen-decomposition path
    dup = gto.M(atom=h2o, basis='sto-3g')
    aux1 = addons.make_auxmol(dup, 'weigend')
    atm, bas, env = gto.conc_env(aux1._atm, aux1._bas, aux1._env, aux1._atm, aux1._bas, aux1._env)

    class _Aux:
        pass
    aux2 = _Aux()
    aux2._atm, aux2._bas, aux2._env = atm, bas, env
    o3 = native.NativeDF(dup, auxmol=aux2).build()
    assert o3.get_naoaux() == aux1.nao_nr(), (o3.get_naoaux(), aux1.nao_nr())      # the duplicate half is linearly dependent
    cd1 = ref.cholesky_eri(dup, aux1)
    got = np.vstack(list(o3.loop()))
    assert np.abs(got.T.dot(got) - cd1.T.dot(cd1)).max() < 1e-7                      # same fitted (pq|rs)
    assert 'torch' not in sys.modules
    stamp('linearly dependent metric done')
    print('NATIVE_ABI_OK', flush=True)


if __name__ == '__main__':
    main()
