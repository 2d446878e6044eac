"""r06 - the SQUARE layout: sq[L][rows][rows] as the ONLY resident copy of the tensor (2x the packed bytes, not
packed + image = 3x).  The J passes read the p >= q runs of the square rows, the K half transform streams them, loop / save / the
`_cderi_dev` property pack on the fly; results must equal the packed layout's and the oracle's (pyscf/df/df_jk.py:280-413,
pyscf/df/df.py:59-72,214-242)."""
import os

import numpy as np
import pytest

from oracle import ref

pytestmark = pytest.mark.gpu


def _obj(cderi, layout, mol=None):
    from pyscf_amd import df
    obj = df.DF(mol)
    obj._cderi = cderi
    obj.layout = layout
    if layout == 'packed':
        obj.k_square = False
    obj.build()
    return obj


def _reference_jk(cderi, dms):
    full = ref.unpack_tril(cderi)
    vj, vk = [], []
    for dm in dms:
        rho = np.einsum('Lpq,pq->L', full, dm)
        vj.append(np.einsum('L,Lpq->pq', rho, full))
        tmp = np.einsum('Lpr,rs->Lps', full, dm)
        vk.append(np.einsum('Lps,Lqs->pq', tmp, full))
    return np.array(vj), np.array(vk)


@pytest.mark.parametrize('nao,naux,nocc', [(130, 301, 33), (257, 96, 161), (61, 17, 1)])
def test_square_layout_equals_packed_and_reference(nao, naux, nocc, tmp_path):
    import torch
    from pyscf_amd import lib
    rng = np.random.default_rng(nao + 1)
    npair = nao * (nao + 1) // 2
    cderi = rng.standard_normal((naux, npair)) / np.sqrt(nao)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c * occ).dot(c.T)
    sq = _obj(cderi, 'square')
    pk = _obj(cderi, 'packed')
    assert sq._layout == 'square' and sq._packed is None and sq._cderi_sq.shape == (naux, (nao + 15) // 16 * 16, (nao + 15) // 16 * 16)
    assert pk._layout == 'packed' and pk._cderi_sq is None
    assert sq.tensor_shape() == (naux, npair) == pk.tensor_shape()
    # MO branch (tagged density: first J pass from the half transform's epilogue), every second-pass schedule
    tag = lib.tag_array(dm, mo_coeff=c, mo_occ=occ)
    vj0, vk0 = _reference_jk(cderi, [dm])
    for pol in ('overlap', 'serial', 'fused', 'auto'):          # 'fused' streams packed rows: the square layout runs 'overlap' instead
        sq.j2_policy = pol
        vj, vk = sq.get_jk(tag, hermi=1)
        assert np.abs(vj - vj0[0]).max() < 1e-10 and np.abs(vk - vk0[0]).max() < 1e-10, pol
    vjp, vkp = pk.get_jk(tag, hermi=1)
    assert np.abs(vjp - vj0[0]).max() < 1e-10 and np.abs(vkp - vk0[0]).max() < 1e-10
    # general-DM branch, three densities, hermi = 0 (two-pass J on the square rows, K with the square rows as second operand)
    dms = rng.standard_normal((3, nao, nao))
    vj0, vk0 = _reference_jk(cderi, dms)
    vj, vk = sq.get_jk(dms, hermi=0)
    assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9
    vj1, _ = sq.get_jk(dms, hermi=0, with_k=False)
    _, vk1 = sq.get_jk(dms, hermi=0, with_j=False)
    assert np.abs(vj1 - vj0).max() < 1e-9 and np.abs(vk1 - vk0).max() < 1e-9
    # symmetric untagged density: eigen-factorised MO branch
    vj2, vk2 = sq.get_jk(dm, hermi=1)
    assert np.abs(vk2 - _reference_jk(cderi, [dm])[1][0]).max() < 1e-10
    # nothing above needed the packed rows
    assert sq._packed is None
    # the boundary format: packed rows out of the square layout, bit for bit what went in
    assert np.array_equal(sq.packed_rows(3, min(9, naux)).cpu().numpy(), cderi[3:min(9, naux)])
    assert np.array_equal(np.vstack(list(sq.loop(blksize=7))), cderi)
    path = str(tmp_path / 'cderi.npy')
    sq.save(path, fmt='npy')
    assert np.array_equal(np.load(path), cderi)
    # the property materialises a packed copy for the consumers that want one (gradients, get_eri, tests)
    assert np.array_equal(sq._cderi_dev.cpu().numpy(), cderi) and sq._layout == 'square'
    eri_sq, eri_pk = sq.get_eri(), pk.get_eri()
    assert np.abs(eri_sq - eri_pk).max() < 1e-10
    sq.to_packed_layout()
    assert sq._layout == 'packed' and sq._cderi_sq is None
    vj, vk = sq.get_jk(tag, hermi=1)
    assert np.abs(vk - vkp).max() < 1e-10 and np.abs(vj - vjp).max() < 1e-10
    del sq, pk
    torch.cuda.empty_cache()


def test_square_layout_built_from_the_integrals_vs_oracle():
    """DF.build() writes the square rows slab by slab (PAMD_cderi_solve into a work slab + PAMD_unpack_tril_slab): rows against the
    oracle's cholesky_eri, the DF-RHF energy of (H2O)_3 cc-pVDZ against the packed layout's and the oracle's SCF, and the analytic
    gradient (which needs the packed tensor: DF.to_packed_layout) against the packed layout's."""
    from pyscf_amd import gto, df, scf
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')
    aux = df.make_auxmol(mol, 'cc-pvdz-jkfit')
    cderi0 = ref.cholesky_eri(mol, aux)
    sq = df.DF(mol, 'cc-pvdz-jkfit')
    sq.layout = 'square'
    sq.build()
    assert sq._layout == 'square' and sq._packed is None
    rows = (mol.nao + 15) // 16 * 16
    assert sq._cderi_sq.shape == (cderi0.shape[0], rows, rows)
    s = sq._cderi_sq.cpu().numpy()
    assert np.abs(s - s.transpose(0, 2, 1)).max() == 0 and np.abs(s[:, mol.nao:, :]).max() == 0 and np.abs(s[:, :, mol.nao:]).max() == 0
    assert np.abs(sq.packed_rows(0, cderi0.shape[0]).cpu().numpy() - cderi0).max() < 1e-10
    # 'auto' picks the square rows for nao >= 128 when the device has the room (it has, at this size)
    au = df.DF(mol, 'cc-pvdz-jkfit').build()
    assert mol.nao < 128 or au._layout == 'square'
    e = {}
    g = {}
    for lay in ('square', 'packed'):
        mf = scf.RHF(mol).density_fit(auxbasis='cc-pvdz-jkfit')
        mf.with_df.layout = lay
        mf.conv_tol = 1e-11
        e[lay] = mf.kernel()
        assert mf.with_df._layout == lay
        g[lay] = mf.nuc_grad_method().kernel()
    assert abs(e['square'] - e['packed']) < 1e-9
    assert np.abs(g['square'] - g['packed']).max() < 1e-8
    e0 = ref.rhf_energy(mol, cderi0) if hasattr(ref, 'rhf_energy') else None
    if e0 is not None:
        assert abs(e['square'] - e0) < 1e-8


def test_layout_budget_keeps_room_for_the_xc_image(monkeypatch):
    """The single HBM budget: with an XC hint larger than what 2x the tensor leaves, 'auto' falls back to the packed rows; what the XC
    plan already holds (lib.hbm) is not asked for twice.  The free-memory figure is pinned (100 GB) so that the decision does not
    depend on what earlier tests of the same process left on the device."""
    import torch
    from pyscf_amd import gto, df
    from pyscf_amd.lib import hbm
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvtz')
    dev = torch.device('cuda', torch.cuda.current_device())
    pinned = 100 << 30
    monkeypatch.setattr(hbm, 'free_bytes', lambda d: pinned)
    kept = hbm.held(dev, 'xc_image')
    hbm.drop(dev, 'xc_image')
    try:
        def make(hint, prefer_image):
            o = df.DF(mol)
            o.xc_image_hint, o.prefer_image = hint, prefer_image
            return o.build()
        # an XC leg that wants everything that is free: neither 3x nor 2x leaves it room -> packed rows (+ what is left)
        assert make(pinned, False)._layout == 'packed' and make(pinned, True)._layout == 'packed'
        hbm.hold(dev, 'xc_image', pinned)           # ... which the plan already holds: nothing left to reserve
        assert make(pinned, False)._layout == 'square'
        hbm.drop(dev, 'xc_image')
        assert make(0, False)._layout == 'square'
        assert make(60 << 30, False)._layout == 'square'     # tensor (tiny) + X block + 4 GB + 60 GB + 12 GB of XC work space <= 100 GB
        # the default preference: packed rows + a FULL image while all three copies fit (the faster second J pass beside the SYRK)
        e = make(0, True)
        assert e._layout == 'packed'
        import numpy as np
        from pyscf_amd import lib
        c = np.linalg.qr(np.random.default_rng(0).standard_normal((mol.nao, mol.nao)))[0]
        occ = np.zeros(mol.nao)
        occ[:mol.nelectron // 2] = 2
        e.get_jk(lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ), hermi=1)
        assert e._cderi_sq is not None and e._cderi_sq.shape[0] == e.tensor_shape()[0]        # the whole image was built
    finally:
        hbm.drop(dev, 'xc_image')
        if kept:
            hbm.hold(dev, 'xc_image', kept)
