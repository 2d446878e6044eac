"""Block-sparse XC path (csrc/xc_sparse.hip, dft/sparse_grid.py): the batched compact-operand kernels through the C ABI
against numpy, and nr_rks / nr_uks on the compact AO subsets against the dense pipeline and the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _setup():
    import torch
    from pyscf_amd import lib
    so = lib.load_library()
    dev = torch.device('cuda', 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return torch, so, dev, st, lib


@pytest.mark.parametrize('G,ncomp,nocc', [(128, 1, 5), (256, 4, 37), (128, 4, 170), (256, 4, 160), (128, 1, 139), (384, 4, 300),
                                           (256, 4, 226), (128, 4, 340)])      # r06: 2 x 128 / 3 x 128 orbital chunks in PAMD_sub_orb_rho (taxol's nocc)
def test_sub_kernels_vs_numpy(G, ncomp, nocc):
    """PAMD_sub_gather_ao / _orb_dot / _scale_ao / _vmat on ragged tiles (ld = 16 ... 272, partial 128-blocks, padding
    columns, an empty tile) against dense numpy."""
    torch, so, dev, st, lib = _setup()
    rng = np.random.default_rng(G + ncomp + nocc)
    nao = 301
    nsubs = [5, 16, 130, 0, 257, 44]
    ntile = len(nsubs)
    lds = [max(16, (n + 15) // 16 * 16) for n in nsubs]
    idx_rows = []
    for n, l in zip(nsubs, lds):
        row = np.full(l, nao, np.int32)
        row[:n] = np.sort(rng.choice(nao, n, replace=False))
        idx_rows.append(row)
    idx = np.concatenate(idx_rows)
    ld = np.array(lds, np.int32)
    idx_off = np.concatenate([[0], np.cumsum(ld)[:-1]]).astype(np.int64)
    ao_off = np.concatenate([[0], np.cumsum(ncomp * G * ld.astype(np.int64))[:-1]]).astype(np.int64)
    aow_off = np.concatenate([[0], np.cumsum(G * ld.astype(np.int64))[:-1]]).astype(np.int64)
    ao_total, aow_total = int((ncomp * G * ld).sum()), int((G * ld).sum())
    ldao = (nao + 15) // 16 * 16
    nvalid = ntile * G - 37                                   # ragged end: the last 37 rows do not exist
    dense = np.zeros((ncomp, ntile * G, ldao))
    dense[:, :nvalid, :nao] = rng.standard_normal((ncomp, nvalid, nao))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_idx, d_ld, d_idx_off, d_ao_off, d_aow_off = t(idx), t(ld), t(idx_off), t(ao_off), t(aow_off)
    d_dense = t(dense)
    ao_c = torch.full((ao_total + 256,), 7.0, dtype=torch.float64, device=dev)
    ao_c[ao_total:].zero_()
    lib.check(so.PAMD_sub_gather_ao(_p(d_dense), C.c_long(ntile * G), ldao, ncomp, C.c_long(0), C.c_long(nvalid), _p(d_ao_off),
                                    _p(d_idx_off), _p(d_ld), _p(d_idx), ntile, G, int(ld.max()), nao, _p(ao_c), st))
    got = ao_c.cpu().numpy()
    want_tiles = []
    for k in range(ntile):
        blk = np.zeros((ncomp, G, lds[k]))
        cols = idx_rows[k] < nao
        blk[:, :, cols] = dense[:, k * G:(k + 1) * G, :][:, :, idx_rows[k][cols]]
        want_tiles.append(blk)
        assert np.array_equal(got[ao_off[k]:ao_off[k] + blk.size].reshape(blk.shape), blk)
    # c = ao . C on the gathered rows of C
    from pyscf_amd.df import df_jk
    cmat = rng.standard_normal((nao, nocc))
    nocc_pad = (nocc + 15) // 16 * 16
    ldo = (nocc_pad + 159) // 160 * 160 if nocc_pad > 160 else nocc_pad
    orb_h = np.zeros(((nao + 16) // 16 * 16, ldo))
    orb_h[:nao, :nocc] = cmat
    orb = t(orb_h)
    npts = ntile * G
    cmo = torch.zeros(ncomp * nocc_pad * npts, dtype=torch.float64, device=dev)
    lib.check(so.PAMD_sub_orb_dot(_p(ao_c), _p(d_ao_off), _p(d_idx_off), _p(d_ld), _p(d_idx), ntile, G, ncomp, _p(orb), ldo,
                                  nocc_pad, _p(cmo), C.c_long(nocc_pad * npts), C.c_long(npts), st))
    want = np.einsum('cgm,mi->cig', dense[:, :, :nao], cmat)           # dense: the dropped columns are simply absent below
    want_sub = np.zeros((ncomp, nocc, npts))
    for k in range(ntile):
        cols = idx_rows[k][idx_rows[k] < nao]
        want_sub[:, :, k * G:(k + 1) * G] = np.einsum('cgm,mi->cig', dense[:, k * G:(k + 1) * G, cols], cmat[cols])
    got = cmo.view(ncomp, nocc_pad, npts)[:, :nocc].cpu().numpy()
    assert np.abs(got - want_sub).max() < 1e-11 * max(1.0, np.abs(want_sub).max())
    if ncomp == 4:
        # r04: the same product with rho / grad rho taken in its epilogue (PAMD_sub_orb_rho), signed "occupations"
        sg = np.where(rng.random(nocc) < 0.3, -1.0, 1.0)
        for sign in (None, sg):
            w = want_sub if sign is None else want_sub * sign[None, :, None]
            rho_want = np.stack([np.einsum('ig,ig->g', w[0], want_sub[0])] +
                                [2 * np.einsum('ig,ig->g', w[0], want_sub[c]) for c in (1, 2, 3)])
            ldg = npts + 64
            rho = torch.full((4, ldg), 3.0 if nocc_pad <= 160 else 0.0, dtype=torch.float64, device=dev)     # several chunks add into zeros
            d_sg = t(sign) if sign is not None else None
            rc = so.PAMD_sub_orb_rho(_p(ao_c), _p(d_ao_off), _p(d_idx_off), _p(d_ld), _p(d_idx), ntile, G, _p(orb), ldo, nocc, nocc_pad,
                                     _p(d_sg) if d_sg is not None else None, _p(rho), C.c_long(ldg), st)
            mt_total = nocc_pad // 16
            nchunk = -(-mt_total // 10)
            if -(-mt_total // nchunk) < 8:                      # fewer than 128 orbitals per 160-chunk
                assert rc == 1                                  # no fused kernel for this shape: the caller runs the two calls
                continue
            assert rc == 0, lib.load_library().PAMD_last_error()
            got_rho = rho[:, :npts].cpu().numpy()
            assert np.abs(got_rho - rho_want).max() < 1e-11 * max(1.0, np.abs(rho_want).max()), (nocc, sign is not None)
    # aow = sum_c wv_c ao_c ; M[idx, idx] += ao0^T aow
    wv = rng.standard_normal((4, npts))
    d_wv = t(wv)
    aow = torch.zeros(aow_total + 256, dtype=torch.float64, device=dev)
    lib.check(so.PAMD_sub_scale_ao(_p(ao_c), _p(d_ao_off), _p(d_aow_off), _p(d_ld), ntile, G, ncomp, int(ld.max()), _p(d_wv),
                                   C.c_long(npts), _p(aow), st))
    work = []
    for k in np.argsort(-ld):
        n128 = -(-int(ld[k]) // 128)
        work += [(k, a, b) for a in range(n128) for b in range(n128)]
    d_work = t(np.asarray(work, np.int32).reshape(-1))
    M = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_sub_vmat(_p(ao_c), _p(d_ao_off), _p(aow), _p(d_aow_off), _p(d_idx_off), _p(d_ld), _p(d_idx), _p(d_work),
                               len(work), G, nao, _p(M), C.c_long(nao), st))
    Mw = np.zeros((nao, nao))
    for k in range(ntile):
        cols = idx_rows[k][idx_rows[k] < nao]
        a = dense[:, k * G:(k + 1) * G, cols]
        w = wv[:ncomp, k * G:(k + 1) * G]
        aw = np.einsum('cg,cgm->gm', w, a)
        if len(cols) == 0:
            continue
        got_aow = aow[aow_off[k]:aow_off[k] + G * lds[k]].view(G, lds[k])[:, :len(cols)].cpu().numpy()
        assert np.abs(got_aow - aw).max() < 1e-12 * max(1.0, np.abs(aw).max())
        Mw[np.ix_(cols, cols)] += a[0].T.dot(aw)
    assert np.abs(M.cpu().numpy() - Mw).max() < 1e-11 * max(1.0, np.abs(Mw).max())
    # r04: V = M + M^T accumulated on its lower triangle by balanced blocks (PAMD_sub_vmat_sym), completed by PAMD_mirror_tril
    so.PAMD_sub_vmat_work.restype = C.c_long
    ld32 = np.ascontiguousarray(ld, dtype=np.int32)
    nw = so.PAMD_sub_vmat_work(ld32.ctypes.data_as(C.c_void_p), ntile, None)
    wl = np.zeros(nw * 6, np.int32)
    assert so.PAMD_sub_vmat_work(ld32.ctypes.data_as(C.c_void_p), ntile, wl.ctypes.data_as(C.c_void_p)) == nw
    items = wl.reshape(-1, 6)
    assert np.all(items[:, 2] <= 8) and np.all(items[:, 4] <= 8) and np.all(items[:, 1] >= items[:, 3])
    for k in range(ntile):                                    # the pieces of a tile cover its groups exactly once
        mine = items[(items[:, 0] == k) & (items[:, 5] == 1)]
        assert mine[:, 2].sum() * 16 == ld[k] and np.array_equal(np.sort(mine[:, 1]), np.concatenate([[0], np.cumsum(mine[np.argsort(mine[:, 1]), 2])[:-1] * 16]))
    L = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_sub_vmat_sym(_p(ao_c), _p(d_ao_off), _p(aow), _p(d_aow_off), _p(d_idx_off), _p(d_ld), _p(d_idx), _p(t(wl)),
                                   int(nw), G, nao, _p(L), C.c_long(nao), st))
    V = torch.empty((nao, nao), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_mirror_tril(_p(L), nao, nao, _p(V), st))
    Vw = Mw + Mw.T
    assert float(torch.triu(L, 1).abs().max()) == 0.0                       # nothing above the diagonal
    assert np.abs(V.cpu().numpy() - Vw).max() < 1e-11 * max(1.0, np.abs(Vw).max())


@pytest.fixture(scope='module')
def water4():
    from pyscf_amd import gto, dft
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(4), basis='cc-pvdz')
    grids = dft.Grids(mol)
    grids.level = 1
    grids.build()
    return mol, grids


@pytest.mark.parametrize('xc', ['lda,vwn', 'b3lyp'])
def test_nr_rks_sparse_equals_dense_and_oracle(water4, xc):
    """nr_rks on compact AO subsets (several tile sizes, cached and recomputed AO image, tagged and untagged - also
    indefinite - densities) against the dense pipeline (1e-10) and the numpy / sympy oracle."""
    from oracle import ref_dft
    from pyscf_amd import dft, lib
    from pyscf_amd.dft import libxc
    mol, grids = water4
    nao, nocc = mol.nao, mol.nelectron // 2
    rng = np.random.default_rng(3)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0] * 0.4
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c[:, :nocc] * 2).dot(c[:, :nocc].T)
    dm_tag = lib.tag_array(dm, mo_coeff=c, mo_occ=occ)
    sym = rng.standard_normal((nao, nao)) * 0.02
    dm_indef = dm + sym + sym.T                       # symmetric, not positive: the signed factorisation
    dense = dft.NumInt()
    dense.sparse = False
    hyb, fac = libxc.parse_xc(xc)
    gga = libxc.xc_type(xc) == 'GGA'
    want = {}
    for name, d in (('tag', dm_tag), ('dm', dm), ('indef', dm_indef)):
        want[name] = dense.nr_rks(mol, grids, xc, d)
    n0, e0, v0 = ref_dft.nr_rks(mol, grids.coords, grids.weights, fac, gga, dm)
    assert abs(want['dm'][0] - n0) < 1e-9 and abs(want['dm'][1] - e0) < 1e-9 and np.abs(want['dm'][2] - v0).max() < 1e-9
    for tile, cache in ((128, True), (256, False), (1024, 'auto')):
        ni = dft.NumInt()
        ni.sparse_tile, ni.ao_cache = tile, cache
        ni.sparse_chunk_points = 4096                 # several chunks
        for name, d in (('tag', dm_tag), ('dm', dm), ('indef', dm_indef)):
            n, e, v = ni.nr_rks(mol, grids, xc, d)
            wn, we, wv = want[name]
            assert abs(n - wn) < 1e-10 and abs(e - we) < 1e-10, (tile, name, n - wn, e - we)
            assert np.abs(v - wv).max() < 1e-10, (tile, name, np.abs(v - wv).max())
        plan = ni.sparse_plan(mol, grids, gga)
        assert 0 < plan.density <= 1 and (plan.ao_c is not None) == bool(cache)
    # two densities in one call
    ni = dft.NumInt()
    n2, e2, v2 = ni.nr_rks(mol, grids, xc, np.array([dm, dm_indef]))
    assert np.abs(v2[0] - want['dm'][2]).max() < 1e-10 and np.abs(v2[1] - want['indef'][2]).max() < 1e-10


def test_nr_uks_sparse_equals_dense(water4):
    from pyscf_amd import dft, lib
    mol, grids = water4
    nao = mol.nao
    na, nb = mol.nelectron // 2 + 1, mol.nelectron // 2 - 1
    rng = np.random.default_rng(5)
    c = np.array([np.linalg.qr(rng.standard_normal((nao, nao)))[0] * 0.4 for _ in range(2)])
    occ = np.zeros((2, nao))
    occ[0, :na] = 1
    occ[1, :nb] = 1
    dms = np.array([c[s][:, occ[s] > 0].dot(c[s][:, occ[s] > 0].T) for s in range(2)])
    dense = dft.NumInt()
    dense.sparse = False
    for xc in ('lda,vwn', 'b3lyp', 'pbe'):
        for d in (lib.tag_array(dms, mo_coeff=c, mo_occ=occ), dms):
            wn, we, wv = dense.nr_uks(mol, grids, xc, d)
            ni = dft.NumInt()
            ni.sparse_tile = 256
            n, e, v = ni.nr_uks(mol, grids, xc, d)
            assert np.abs(n - wn).max() < 1e-10 and abs(e - we) < 1e-10 and np.abs(v - wv).max() < 1e-10, xc


def test_sparse_plan_follows_the_molecule(water4):
    """Plans and shell tables are keyed on content: a new geometry with the same sizes gets its own plan."""
    from pyscf_amd import gto, dft
    from pyscf_amd.data import clusters
    mol, grids = water4
    ni = dft.NumInt()
    atoms = [(s, (x * 1.02, y, z)) for s, (x, y, z) in clusters.water_cluster(4)]
    mol2 = gto.M(atom=atoms, basis='cc-pvdz')
    g2 = dft.Grids(mol2)
    g2.level = 1
    g2.build()
    dm = np.eye(mol.nao) * 0.1
    e1 = ni.nr_rks(mol, grids, 'lda,vwn', dm)[1]
    e2 = ni.nr_rks(mol2, g2, 'lda,vwn', dm)[1]
    ref = dft.NumInt()
    ref.sparse = False
    assert abs(e2 - ref.nr_rks(mol2, g2, 'lda,vwn', dm)[1]) < 1e-10 and abs(e1 - e2) > 1e-6


@pytest.mark.parametrize('xc', ['lda,vwn', 'b3lyp'])
def test_fxc_sparse_equals_dense(water4, xc):
    """nr_rks_fxc / nr_uks_fxc / nr_rks_fxc_st on the compact AO subsets - first-order densities eigen-factorised, or taken
    from the factors the response solvers tag them with (D = L R^T [+ h.c.]) - against the dense pipeline, 1e-10."""
    from pyscf_amd import dft, lib
    mol, grids = water4
    nao, nocc = mol.nao, mol.nelectron // 2
    rng = np.random.default_rng(8)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0] * 0.4
    occ = np.zeros(nao)
    occ[:nocc] = 2
    co, cv = c[:, :nocc], c[:, nocc:]
    dm0 = (co * 2).dot(co.T)
    xs = rng.standard_normal((2, nocc, nao - nocc)) * 0.05
    rights = [2 * cv.dot(x.T) for x in xs]
    dm1 = np.array([co.dot(r.T) for r in rights])                      # non-symmetric trial densities
    dense = dft.NumInt()
    dense.sparse = False
    ni = dft.NumInt()
    ni.sparse_tile = 256
    ni.sparse_chunk_points = 8192
    want = dense.nr_rks_fxc(mol, grids, xc, dm0, dm1, hermi=0)
    got_plain = ni.nr_rks_fxc(mol, grids, xc, dm0, dm1, hermi=0)
    got_tag = ni.nr_rks_fxc(mol, grids, xc, lib.tag_array(dm0, mo_coeff=c, mo_occ=occ),
                            lib.tag_array(dm1, lowrank=([co, co], rights, False)), hermi=0)
    assert np.abs(got_plain - want).max() < 1e-10 and np.abs(got_tag - want).max() < 1e-10
    dsym = dm1 + dm1.transpose(0, 2, 1)
    want = dense.nr_rks_fxc(mol, grids, xc, dm0, dsym, hermi=1)
    got = ni.nr_rks_fxc(mol, grids, xc, dm0, lib.tag_array(dsym, lowrank=([co, co], rights, True)), hermi=1)
    assert np.abs(got - want).max() < 1e-10
    for singlet in (True, False):
        want = dense.nr_rks_fxc_st(mol, grids, xc, dm0, dm1 * .5, singlet=singlet)
        got = ni.nr_rks_fxc_st(mol, grids, xc, lib.tag_array(dm0, mo_coeff=c, mo_occ=occ),
                               lib.tag_array(dm1 * .5, lowrank=([co, co], [r * .5 for r in rights], False)), singlet=singlet)
        got2 = ni.nr_rks_fxc_st(mol, grids, xc, dm0, dm1 * .5, singlet=singlet)
        assert np.abs(got - want).max() < 1e-10 and np.abs(got2 - want).max() < 1e-10, singlet
    # open shell
    dm0u = np.array([(co[:, :nocc] ).dot(co[:, :nocc].T), (co[:, :nocc - 2]).dot(co[:, :nocc - 2].T)])
    d1u = np.array([dm1[0] * .5, dm1[1] * .3])
    want = dense.nr_uks_fxc(mol, grids, xc, dm0u, d1u)
    got = ni.nr_uks_fxc(mol, grids, xc, dm0u, d1u)
    assert np.abs(got - want).max() < 1e-10
