"""Worker of tests/test_gpu_native_r04.py: the host-array XC entry points (PAMD_grid_weights_host, PAMD_xc_create, PAMD_xc_nr_rks,
PAMD_xc_nr_uks) from numpy + ctypes in a process that never imports torch.

  * NativeGrids vs the oracle's numpy Becke partition (oracle/ref_dft.build_grids), coordinates and weights
  * NativeNumInt.nr_rks / nr_uks vs the oracle's dense numpy restatement (sympy functionals), tagged / untagged / indefinite
    densities, several densities per call, LDA and GGA
  * the reference's own DF-RKS golden -76.690346887915879 (pyscf/dft/test/test_h2o.py:236-240: H2O / 6-31G, B88,VWN, 'weigend',
    Treutler pruning, (50, 194) grids) through the stock RKS driver of this package with with_df = NativeDF, _numint =
    NativeNumInt, grids = NativeGrids - the one-electron integrals come from the oracle (in a real PySCF process: libcint).
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
    from oracle import ref, ref_dft
    from pyscf_amd import gto, dft, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df.native import NativeDF
    from pyscf_amd.dft import libxc, radi, gen_grid
    from pyscf_amd.dft.native import NativeGrids, NativeNumInt
    assert 'torch' not in sys.modules

    # ---- grids: host tables + PAMD_grid_weights_host vs the numpy partition of the oracle
    mol = gto.M(atom=clusters.water_cluster(2), basis='cc-pvdz')
    grids = NativeGrids(mol)
    grids.level = 1
    grids.build()
    c0, w0 = ref_dft.build_grids(mol, level=1)
    assert grids.coords.shape == c0.shape and np.abs(grids.coords - c0).max() < 1e-12
    assert np.abs(grids.weights - w0).max() < 1e-11 * np.abs(w0).max(), np.abs(grids.weights - w0).max()
    for scheme, name in ((gen_grid.stratmann, 'stratmann'), (gen_grid.becke_lko, 'lko')):
        g2 = NativeGrids(mol)
        g2.level = 0
        g2.becke_scheme = scheme
        g2.build()
        c1, w1 = ref_dft.build_grids(mol, level=0, scheme=name)
        assert np.abs(g2.weights - w1).max() < 1e-11 * np.abs(w1).max(), name
    stamp('grids')

    # ---- nr_rks / nr_uks vs the oracle
    nao, nocc = mol.nao, mol.nelectron // 2
    s = ref.int1e(mol, 'ovlp')
    rng = np.random.RandomState(7)
    x = rng.rand(nao, nao)
    w, v = np.linalg.eigh(x.T.dot(s).dot(x))
    c = x.dot(v / np.sqrt(w)).dot(v.T)                      # S-orthonormal orbitals
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c * occ).dot(c.T)
    ni = NativeNumInt()
    for xc in ('lda,vwn', 'b3lyp', 'pbe'):
        hyb, fac = libxc.parse_xc(xc)
        gga = libxc.xc_type(xc) == 'GGA'
        n0, e0, v0 = ref_dft.nr_rks(mol, grids.coords, grids.weights, fac, gga, dm)
        for tag, d in (('tagged', lib.tag_array(dm, mo_coeff=c, mo_occ=occ)), ('untagged', dm)):
            n1, e1, v1 = ni.nr_rks(mol, grids, xc, d)
            assert abs(n1 - n0) < 1e-10 * abs(n0) and abs(e1 - e0) < 1e-10 * abs(e0), (xc, tag, n1 - n0, e1 - e0)
            assert np.abs(v1 - v0).max() < 1e-9 * max(1.0, np.abs(v0).max()), (xc, tag, np.abs(v1 - v0).max())
    assert abs(n1 - mol.nelectron) < 2e-2
    info = ni.plan_info(mol, grids, 'b3lyp')
    assert info['tiles'] == -(-grids.size // 512) and 0 < info['density'] <= 1, info
    # several densities in one call, one of them indefinite (a difference density)
    dm2 = dm - 1.5 * (c[:, :3] * 2).dot(c[:, :3].T) + 0.3 * (c[:, nocc:nocc + 2]).dot(c[:, nocc:nocc + 2].T)
    hyb, fac = libxc.parse_xc('b3lyp')
    ns, es, vs = ni.nr_rks(mol, grids, 'b3lyp', np.array([dm, dm2]))
    for k, d in enumerate((dm, dm2)):
        n0, e0, v0 = ref_dft.nr_rks(mol, grids.coords, grids.weights, fac, True, d)
        assert abs(ns[k] - n0) < 1e-9 and abs(es[k] - e0) < 1e-9 and np.abs(vs[k] - v0).max() < 1e-8, k
    # hybrid-only: nothing to integrate
    n, e, vv = ni.nr_rks(mol, grids, 'hf', dm)
    assert n == 0 and e == 0 and not vv.any()
    # spin-polarised
    occa, occb = np.zeros(nao), np.zeros(nao)
    occa[:nocc] = 1
    occb[:nocc - 1] = 1
    dma, dmb = (c * occa).dot(c.T), (c * occb).dot(c.T)
    for xc in ('lda,vwn', 'b3lyp'):
        hyb, fac = libxc.parse_xc(xc)
        gga = libxc.xc_type(xc) == 'GGA'
        n0, e0, v0 = ref_dft.nr_uks(mol, grids.coords, grids.weights, fac, gga, dma, dmb)
        for d in (np.array([dma, dmb]), lib.tag_array(np.array([dma, dmb]), mo_coeff=np.array([c, c]), mo_occ=np.array([occa, occb]))):
            n1, e1, v1 = ni.nr_uks(mol, grids, xc, d)
            assert np.abs(np.asarray(n1) - np.asarray(n0)).max() < 1e-9 and abs(e1 - e0) < 1e-9 * abs(e0), (xc, n1, n0, e1, e0)
            assert np.abs(v1 - v0).max() < 1e-8 * max(1.0, np.abs(v0).max()), (xc, np.abs(v1 - v0).max())
    ni.reset()
    # the same over a device list (tiles dealt round-robin over the parts; the test box has one GPU: the list repeats it)
    for devs in ([0, 0], [0, 0, 0, 0, 0]):
        nm = NativeNumInt(devices=devs)
        for xc in ('lda,vwn', 'b3lyp'):
            hyb, fac = libxc.parse_xc(xc)
            gga = libxc.xc_type(xc) == 'GGA'
            n0, e0, v0 = ref_dft.nr_rks(mol, grids.coords, grids.weights, fac, gga, dm)
            n1, e1, v1 = nm.nr_rks(mol, grids, xc, lib.tag_array(dm, mo_coeff=c, mo_occ=occ))
            assert abs(n1 - n0) < 1e-10 * abs(n0) and abs(e1 - e0) < 1e-10 * abs(e0), (devs, xc)
            assert np.abs(v1 - v0).max() < 1e-9 * max(1.0, np.abs(v0).max()), (devs, xc, np.abs(v1 - v0).max())
        hyb, fac = libxc.parse_xc('b3lyp')
        n0, e0, v0 = ref_dft.nr_uks(mol, grids.coords, grids.weights, fac, True, dma, dmb)
        n1, e1, v1 = nm.nr_uks(mol, grids, 'b3lyp', np.array([dma, dmb]))
        assert np.abs(np.asarray(n1) - np.asarray(n0)).max() < 1e-9 and abs(e1 - e0) < 1e-9 * abs(e0) and np.abs(v1 - v0).max() < 1e-8
        assert nm.plan_info(mol, grids, 'b3lyp')['tiles'] >= -(-grids.size // 512)
        nm.reset()
    stamp('nr_rks / nr_uks vs oracle')

    # ---- the reference's DF-RKS golden through the stock driver with the three native objects
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False            # test_h2o.py:86-89
    try:
        h2o = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='6-31g')
        mf = dft.RKS(h2o)
        mf.xc = 'b88, vwn'
        mf.with_df = NativeDF(h2o, auxbasis='weigend')
        mf._numint = NativeNumInt()
        mf.grids = NativeGrids(h2o)
        mf.grids.prune = gen_grid.treutler_prune
        mf.grids.atom_grid = {'H': (50, 194), 'O': (50, 194)}
        h1 = ref.int1e(h2o, 'kin') + ref.int1e(h2o, 'nuc')
        s1 = ref.int1e(h2o, 'ovlp')
        mf.get_hcore = lambda mol=None: h1
        mf.get_ovlp = lambda mol=None: s1
        mf.init_guess = '1e'
        mf.device_scf = False
        mf.conv_tol = 1e-10
        e = mf.kernel()
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    print('DF-RKS B88,VWN / weigend through NativeDF + NativeNumInt + NativeGrids: E = %.12f (%d cycles)' % (e, mf.cycles), flush=True)
    assert mf.converged and abs(e - -76.690346887915879) < 1e-8, e
    assert 'torch' not in sys.modules
    stamp('DF-RKS golden')
    print('NATIVE_XC_OK', flush=True)


if __name__ == '__main__':
    main()
