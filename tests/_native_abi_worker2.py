"""Worker of tests/test_gpu_native_r04.py: the r04 additions to the host-array C ABI, from plain numpy + ctypes in a process that
never imports torch:

  * PAMD_df_create_multi - the aux index sharded over a device LIST inside one process (SURVEY.md 8(b) mi_ctx_create; the serial
    L-block loop it replaces: pyscf/df/df_jk.py:362-381).  The test box has one GPU, so the list repeats device 0: sharding, one
    host thread per part, the gather of the partial [J~ | K] (device-to-device and, with PAMD_DF_PEER=0, the host bounce) and
    the fixed-order sum all run; results against the oracle to 1e-11 and against the reference's goldens
    (pyscf/df/test/test_df_jk.py:144-156).
  * rows that do not fit a device-memory cap live in page-locked host memory and are streamed under the kernels (the out-of-core
    twin, pyscf/df/outcore.py:109-232): same numbers, layout says how many rows went where.
  * omega: long-range / short-range tensors (DF.range_coulomb, pyscf/df/df.py:298-333) against the oracle's attenuated integrals.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    t0 = time.perf_counter()

    def stamp(what):
        print('[%7.2f s] %s' % (time.perf_counter() - t0, what), flush=True)
    from oracle import ref
    from pyscf_amd import gto, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df import addons, native
    assert 'torch' not in sys.modules
    h2o = 'O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587'
    mol = gto.M(atom=h2o, basis='cc-pvdz')
    nao = mol.nao
    np.random.seed(1)
    dms = np.random.random((2, nao, nao))
    cderi = ref.cholesky_eri(mol, addons.make_auxmol(mol, 'weigend'))
    vj0, vk0 = ref.get_jk(cderi, dms, hermi=0)

    # ---- several parts: G4 goldens, oracle 1e-11, row export across part boundaries, layout
    for devs in ([0], [0, 0], [0, 0, 0]):
        obj = native.NativeDF(mol, auxbasis='weigend', devices=devs).build()
        lay = obj.layout()
        assert lay['parts'] == len(devs) and sum(lay['part_rows']) == 71 == obj.get_naoaux(), lay
        assert max(lay['part_rows']) - min(lay['part_rows']) <= 1 and lay['rows_host'] == 0, lay
        vj, vk = obj.get_jk(dms, hermi=0)
        assert abs(lib.fp(vj) - -194.15910890730066) < 1e-9 and abs(lib.fp(vk) - -46.365071587653517) < 1e-9, devs
        assert np.abs(vj - vj0).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11, (devs, np.abs(vj - vj0).max(), np.abs(vk - vk0).max())
        vj1, none = obj.get_jk(dms, hermi=0, with_k=False)
        none2, vk1 = obj.get_jk(dms, hermi=0, with_j=False)
        assert none is None and none2 is None and np.abs(vj1 - vj0).max() < 1e-11 and np.abs(vk1 - vk0).max() < 1e-11
        got = np.vstack(list(obj.loop(20)))
        assert got.shape == cderi.shape and np.abs(got - cderi).max() < 1e-10, devs
        obj.reset()
    stamp('multi-part handles: goldens, oracle, export')

    # ---- MO branch over parts (fused J: promise kept / broken), UHF-style sets, the host-bounce gather
    m2 = gto.M(atom=clusters.water_cluster(2), basis='cc-pvtz')
    cd = ref.cholesky_eri(m2, addons.make_auxmol(m2))
    n, nocc = m2.nao, m2.nelectron // 2
    c = np.linalg.qr(np.random.RandomState(3).rand(n, n))[0]
    occ = np.zeros(n)
    occ[:nocc] = 2
    dm = (c[:, :nocc] * 2).dot(c[:, :nocc].T)
    vjm, vkm = ref.get_jk(cd, dm, 1, mo_coeff=c, mo_occ=occ)
    dm_other = dm + 0.01 * np.eye(n)
    vjo = ref.get_jk(cd, dm_other, 1)[0]
    dms2 = np.array([(c[:, :nocc]).dot(c[:, :nocc].T), (c[:, :nocc - 1]).dot(c[:, :nocc - 1].T)])
    occ_a, occ_b = np.zeros(n), np.zeros(n)
    occ_a[:nocc] = 1
    occ_b[:nocc - 1] = 1
    vju, vku = ref.get_jk(cd, dms2, 1)

    def mo_cases(o, tag, tol=1e-11):
        vj, vk = o.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
        assert np.abs(vj - vjm).max() < tol and np.abs(vk - vkm).max() < tol, (tag, np.abs(vj - vjm).max(), np.abs(vk - vkm).max())
        vj, vk = o.get_jk(lib.tag_array(dm_other, mo_coeff=c, mo_occ=occ), hermi=1)      # tag inconsistent with the matrix
        assert np.abs(vj - vjo).max() < tol and np.abs(vk - vkm).max() < tol, tag
        vj, vk = o.get_jk(lib.tag_array(dms2, mo_coeff=np.array([c, c]), mo_occ=np.array([occ_a, occ_b])), hermi=1)
        assert np.abs(vj - vju).max() < tol and np.abs(vk - vku).max() < tol, tag
        vj, vk = o.get_jk(dm_other + 0.03 * np.random.RandomState(8).rand(n, n), hermi=0)       # general branch
        w = ref.get_jk(cd, dm_other + 0.03 * np.random.RandomState(8).rand(n, n), 0)
        assert np.abs(vj - w[0]).max() < tol and np.abs(vk - w[1]).max() < tol, tag

    o2 = native.NativeDF(m2, devices=[0, 0, 0]).build()
    mo_cases(o2, 'three parts')
    o2.reset()
    os.environ['PAMD_DF_PEER'] = '0'
    try:
        o2 = native.NativeDF(m2, devices=[0, 0]).build()
        assert o2.layout()['peer'] == 0
        mo_cases(o2, 'two parts, host bounce')
        o2.reset()
    finally:
        del os.environ['PAMD_DF_PEER']
    stamp('MO branch over parts')

    # ---- out of core: a cap on the device memory leaves rows in page-locked host memory, streamed block by block
    npair = n * (n + 1) // 2
    naux2 = cd.shape[0]
    for cap_rows, devs in ((naux2 // 3, None), (naux2 // 5, [0, 0])):
        # cap sized in tensor rows: 1/4 of the cap holds resident rows (csrc/df_handle.hip: build_rows)
        cap = int(cap_rows * npair * 8 * 4)
        o3 = native.NativeDF(m2, devices=devs, max_device_bytes=cap).build()
        lay = o3.layout()
        assert lay['rows_host'] > 0 and lay['rows_resident'] + lay['rows_host'] == naux2, lay
        print('cap %.1f MB -> layout %s' % (cap * 1e-6, lay), flush=True)
        mo_cases(o3, 'streamed %s' % (devs,))
        got = np.vstack(list(o3.loop(37)))
        assert np.abs(got - cd).max() < 1e-10
        o3.reset()
    os.environ['PAMD_DF_DEVICE_BYTES'] = str(int(8 * npair * 8 * 4))      # the environment form of the same cap: 8 resident rows
    try:
        o3 = native.NativeDF(m2).build()
        lay = o3.layout()
        assert lay['rows_host'] >= naux2 - 9, lay
        mo_cases(o3, 'streamed, env cap')
        o3.reset()
    finally:
        del os.environ['PAMD_DF_DEVICE_BYTES']
    stamp('host-resident rows streamed')

    # ---- omega: long-range and short-range tensors against the oracle's attenuated integrals
    mh = gto.M(atom='H 0 0 0; F 0 0 0.92', basis='cc-pvdz')
    aux = addons.make_auxmol(mh)
    nh = mh.nao
    dmh = np.random.RandomState(4).rand(nh, nh)
    dmh = dmh + dmh.T
    o4 = native.NativeDF(mh, devices=[0, 0])
    for omega in (0.3, -0.3, 1.0):
        b = ref.cholesky_eri(mh, aux, omega=omega)          # Cholesky, or eigen-decomposition when the metric is linearly dependent
        vjr, vkr = ref.get_jk(b, dmh, 1)
        vj, vk = o4.get_jk(dmh, hermi=1, omega=omega)
        assert np.abs(vj - vjr).max() < 1e-7 and np.abs(vk - vkr).max() < 1e-7, (omega, np.abs(vj - vjr).max(), np.abs(vk - vkr).max())
    vj, vk = o4.get_jk(dmh, hermi=1)                                      # and the Coulomb tensor of the same object
    cdh = ref.cholesky_eri(mh, aux)
    w0 = ref.get_jk(cdh, dmh, 1)
    assert np.abs(vj - w0[0]).max() < 1e-10 and np.abs(vk - w0[1]).max() < 1e-10
    o4.reset()
    assert 'torch' not in sys.modules
    stamp('range-separated tensors')
    print('NATIVE_R04_OK', flush=True)


if __name__ == '__main__':
    main()
