"""GPU tests at the BASELINE.json sizes through size-independent properties, and config 2."""
import numpy as np
import pytest

from oracle import ref

pytestmark = pytest.mark.gpu


def test_config2_benzene_df_rhf_energy_vs_oracle():
    """BASELINE config 2: benzene cc-pVTZ DF-RHF (nao 264, naux 654), energy vs the CPU oracle to 1e-8 Eh."""
    from pyscf_amd import gto, scf, df
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.BENZENE, basis='cc-pvtz')
    mf = scf.RHF(mol).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and (mol.nao, mf.with_df.get_naoaux()) == (264, 654)
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))

    def veff(dm, c, occ):
        vj, vk = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
        return vj - .5 * vk
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
    assert conv and abs(e - e0) < 1e-8, (e, e0)
    # cderi itself, row for row
    got = np.vstack(list(mf.with_df.loop()))
    assert np.abs(got - cderi).max() < 1e-9


@pytest.fixture(scope='module')
def h2o32():
    import torch
    from pyscf_amd import gto, df
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
    obj = df.DF(mol)
    obj.prefer_image = False          # the SQUARE layout at full size (r06); the packed layouts are exercised on the same rows below
    obj.build()
    torch.cuda.synchronize()
    yield mol, obj
    obj.reset()                      # 61 GB tensor + 123 GB square image back to the allocator for the next module
    torch.cuda.empty_cache()


def test_config3_shard_sum_linearity_symmetry(h2o32):
    """(H2O)_32 cc-pVTZ (nao 1856, naux 4448, 61 GB tensor): J/K of two aux-row shards add up to the
    full result (the multi-GPU decomposition), J/K are linear in D and symmetric."""
    import torch
    from pyscf_amd import df
    from pyscf_amd.df import df_jk
    mol, obj = h2o32
    nao, naux = mol.nao, obj.get_naoaux()
    assert (nao, naux) == (1856, 4448) and obj._cderi_dev.shape == (4448, nao * (nao + 1) // 2)
    dev = obj._cderi_dev.device
    rng = np.random.default_rng(0)
    c = np.linalg.qr(rng.standard_normal((nao, 200)))[0]
    orbs = [df_jk.pad_orbitals(c[:, :160] * np.sqrt(2.0), dev), df_jk.pad_orbitals(c[:, 40:200] * np.sqrt(2.0), dev)]
    dms = [torch.from_numpy((c[:, :160] * 2).dot(c[:, :160].T)[None]).to(dev),
           torch.from_numpy((c[:, 40:200] * 2).dot(c[:, 40:200].T)[None]).to(dev)]
    vj0, vk0 = df_jk.get_jk_device(obj, dms[0], [orbs[0]])
    vj1, vk1 = df_jk.get_jk_device(obj, dms[1], [orbs[1]])
    # symmetry of K, positivity of the Coulomb/exchange energies
    assert float((vk0[0] - vk0[0].T).abs().max()) < 1e-10
    assert float((dms[0][0] * vk0[0]).sum()) > 0
    # shard sum: rows [0, h) + rows [h, naux)
    h = 2224
    parts = []
    for sl in (slice(0, h), slice(h, naux)):
        sub = df.DF(mol)
        sub._cderi_dev = obj._cderi_dev[sl]
        parts.append(df_jk.get_jk_device(sub, dms[0], [orbs[0]]))
    assert float((parts[0][0] + parts[1][0] - vj0).abs().max()) < 1e-9 * float(vj0.abs().max())
    assert float((parts[0][1] + parts[1][1] - vk0).abs().max()) < 1e-9 * float(vk0.abs().max())
    # linearity in D (general-DM branch on the sum of the two densities vs the sum of MO-branch results)
    small = df.DF(mol)
    small._cderi_dev = obj._cderi_dev[:64]
    a = df_jk.get_jk_device(small, dms[0], [orbs[0]])
    b = df_jk.get_jk_device(small, dms[1], [orbs[1]])
    ab = df_jk.get_jk_device(small, dms[0] + dms[1], None)
    assert float((a[0] + b[0] - ab[0]).abs().max()) < 1e-9 * float(ab[0].abs().max())
    assert float((a[1] + b[1] - ab[1]).abs().max()) < 1e-9 * float(ab[1].abs().max())


def test_config3_tensor_vs_subcluster_oracle(h2o32):
    """Checks the 61 GB on-device tensor against CPU integrals without building it on the CPU: for the
    AO pairs of the first two waters, the fitted diagonal integrals (pq|pq)_DF = sum_L B[L,pq]^2 from the
    full-cluster tensor must be >= those of an oracle DF build of the 2-water sub-cluster (enlarging the
    aux space can only increase the fitted self-repulsion - variational property of density fitting) and
    close to them."""
    from pyscf_amd import gto, df
    mol, obj = h2o32
    from pyscf_amd.data import clusters
    sub = gto.M(atom=clusters.water_cluster(32)[:6], basis='cc-pvtz')
    aux = df.make_auxmol(sub)
    cd_sub = ref.cholesky_eri(sub, aux)                      # (278, 6786)
    nsub = sub.nao
    npair_sub = nsub * (nsub + 1) // 2
    cols = obj._cderi_dev[:, :npair_sub].cpu().numpy()       # the same AO pairs in the big tensor
    d_big = np.einsum('Lp,Lp->p', cols, cols)
    d_sub = np.einsum('Lp,Lp->p', cd_sub, cd_sub)
    assert np.all(d_big > -1e-12)
    assert np.all(d_big >= d_sub - 1e-9)                      # variational property of density fitting
    assert np.abs(d_big - d_sub).max() < 5e-3                 # and the two fits are close


def test_water4_tz_screened_tensor_and_energy_vs_oracle():
    """(H2O)_4 cc-pVTZ: inter-molecular shell pairs exercise the primitive-pair screening of the device
    pair tables (EXPCUTOFF = 60, as libcint's default) against the unscreened oracle: tensor 1e-9,
    DF-RHF energy 1e-8 Eh."""
    from pyscf_amd import gto, scf, df
    from pyscf_amd.data import clusters
    from pyscf_amd.gto.moleintor import get_engine
    import torch
    mol = gto.M(atom=clusters.water_cluster(4), basis='cc-pvtz')
    aux = df.make_auxmol(mol)
    eng = get_engine(mol, aux, torch.device('cuda', 0))
    nsh = eng.ao.n
    assert sum(pc.n for pc in eng.pair_classes()) <= nsh * (nsh + 1) // 2
    mf = scf.RHF(mol).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    cderi = ref.cholesky_eri(mol, aux)
    got = np.vstack(list(mf.with_df.loop()))
    assert np.abs(got - cderi).max() < 1e-9

    def veff(dm, c, occ):
        vj, vk = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
        return vj - .5 * vk
    conv, e0 = ref.rhf_kernel(mol, veff, conv_tol=1e-10)[:2]
    assert mf.converged and conv and abs(e - e0) < 1e-8, (e, e0)


# ------------------------------------------------------------------------------------------------------------------
# Row N1: parity AT the target size.  The golden numbers come from the CPU oracle alone (its own McMurchie-Davidson
# tensor, streamed from disk; tools/gen_golden_fullsize.py) and are committed under tests/golden/.
def _golden(name):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name)
    if not os.path.exists(path):
        pytest.skip('%s not generated (tools/gen_golden_fullsize.py)' % name)
    with open(path) as f:
        return json.load(f)


def test_config3_tensor_columns_vs_oracle(h2o32):
    """The 61 GB device tensor against oracle integrals, element-wise, on a seeded sample of AO shells spread over the
    cluster (every packed column pq of those rows p, all 4448 aux rows): raw (Q|pq) from the oracle, the oracle's own
    Cholesky factor of (P|Q) applied by trsm (pyscf/df/incore.py:129-220), compared with cderi[:, cols] to 1e-9.
    Exercises the primitive-pair screening and the explicit inverse factor at naux = 4448."""
    import scipy.linalg
    from pyscf_amd import df
    mol, obj = h2o32
    aux = obj.auxmol
    low = scipy.linalg.cholesky(ref.int2c2e(aux), lower=True)
    loc = ref.ao_loc(mol)
    rng = np.random.RandomState(5)
    shells = sorted(set(rng.randint(0, mol.nbas, size=6).tolist()) | {0, mol.nbas - 1})
    ncols = 0
    worst = 0.0
    for ish in shells:
        raw = ref.int3c2e_slab(mol, aux, ish, ish + 1)
        want = scipy.linalg.solve_triangular(low, raw, lower=True, overwrite_b=True, check_finite=False)
        p0, p1 = int(loc[ish]), int(loc[ish + 1])
        pq0, pq1 = p0 * (p0 + 1) // 2, p1 * (p1 + 1) // 2
        got = obj._cderi_dev[:, pq0:pq1].cpu().numpy()
        worst = max(worst, float(np.abs(got - want).max()))
        ncols += pq1 - pq0
    assert ncols > 200
    assert worst < 1e-9, worst


def test_config3_full_jk_vs_oracle_golden(h2o32):
    """One full-size J/K build (all 4448 aux rows, MO branch on the square image AND on the packed operand, and the
    general-DM branch on a row subset) against J/K computed by the oracle from ITS OWN tensor: 4096 sampled entries,
    Frobenius norms, traces and lib.fp fingerprints, 1e-9 relative (pyscf/df/test/test_df_jk.py:57-59 style)."""
    import torch
    from oracle import golden_util
    from pyscf_amd import lib
    from pyscf_amd.df import df_jk
    g = _golden('h2o32_ccpvtz_oracle.json')
    if 'vj_sample' not in g:
        pytest.skip('J/K golden not generated yet')
    mol, obj = h2o32
    nao, nocc = mol.nao, mol.nelectron // 2
    assert (g['nao'], g['naux'], g['nocc']) == (nao, obj.get_naoaux(), nocc)
    dev = obj._cderi_dev.device
    c = golden_util.synthetic_orbitals(nao, nocc)
    dm = 2 * c.dot(c.T)
    ri, ci = golden_util.sample_positions(nao, 4096)
    vj_s, vk_s = np.array(g['vj_sample']), np.array(g['vk_sample'])
    dms = torch.from_numpy(dm[None]).to(dev)
    orbs = [df_jk.pad_orbitals(c * np.sqrt(2.0), dev)]

    def check(vjt, vkd, tag):
        vj = lib.unpack_tril(vjt.cpu().numpy(), 1)[0]
        vk = vkd.cpu().numpy()[0]
        assert np.abs(vj[ri, ci] - vj_s).max() < 1e-9 * g['vj_absmax'], tag
        assert np.abs(vk[ri, ci] - vk_s).max() < 1e-9 * g['vk_absmax'], tag
        assert abs(np.linalg.norm(vj) - g['vj_norm']) < 1e-9 * g['vj_norm'], tag
        assert abs(np.linalg.norm(vk) - g['vk_norm']) < 1e-9 * g['vk_norm'], tag
        assert abs(np.einsum('ij,ji', dm, vj) - g['tr_d_vj']) < 1e-9 * abs(g['tr_d_vj']), tag
        assert abs(np.einsum('ij,ji', dm, vk) - g['tr_d_vk']) < 1e-9 * abs(g['tr_d_vk']), tag
        assert abs(golden_util.fp(vj) - g['vj_fp']) < 1e-9 * g['vj_norm'], tag
        assert abs(golden_util.fp(vk) - g['vk_fp']) < 1e-9 * g['vk_norm'], tag
    check(*df_jk.get_jk_device(obj, dms, orbs), 'square layout')
    assert obj._layout == 'square'          # r06: the square rows are what build() made (the packed copy the earlier test asked for is lazy)
    for pol in ('serial', 'overlap'):
        obj.j2_policy = pol
        check(*df_jk.get_jk_device(obj, dms, orbs), 'square layout, second J pass ' + pol)
    obj.j2_policy = 'auto'
    # the packed layout on the same rows (ranks without the HBM): packed-operand half transform with / without the diagonal-block image
    from pyscf_amd import df as _df
    pk = _df.DF(obj.mol)
    pk._cderi_dev = obj.packed_rows(0, obj.tensor_shape()[0])
    pk._naux, pk.auxmol = obj._naux, obj.auxmol
    pk.k_square = False
    try:
        check(*df_jk.get_jk_device(pk, dms, orbs), 'packed operand + diagonal-block image')
        assert pk._cderi_diag is not None and pk._cderi_diag.shape[0] == obj.get_naoaux()
        pk._cderi_diag, pk.k_diag = None, False
        check(*df_jk.get_jk_device(pk, dms, orbs), 'packed operand')
        pk.j2_policy = 'fused'
        check(*df_jk.get_jk_device(pk, dms, orbs), 'packed operand, second J pass inside the SYRK')
    finally:
        del pk
        torch.cuda.empty_cache()
    # the reference-style host API (numpy in / out, tagged DM) on the same tensor
    vj_h, vk_h = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=np.full(nocc, 2.0)), hermi=1)
    assert np.abs(vj_h[ri, ci] - vj_s).max() < 1e-9 * g['vj_absmax']
    assert np.abs(vk_h[ri, ci] - vk_s).max() < 1e-9 * g['vk_absmax']


@pytest.mark.parametrize('xc', ['', 'b3lyp'])
def test_water8_tz_converged_energy_vs_oracle_golden(xc):
    """(H2O)_8 cc-pVTZ (nao 464, naux 1112): converged DF-RHF and DF-RKS B3LYP total energies against the oracle's own
    SCF (golden file), 1e-8 Eh - the north-star gate, as pyscf/dft/test/test_h2o.py:236-240 does for one water."""
    from pyscf_amd import gto, scf, dft
    from pyscf_amd.data import clusters
    g = _golden('h2o8_ccpvtz_oracle.json')
    key = 'e_rks_' + xc if xc else 'e_rhf'
    if key not in g:
        pytest.skip(key + ' not in the golden file')
    mol = gto.M(atom=clusters.water_cluster(8), basis='cc-pvtz')
    mf = (dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and (mol.nao, mf.with_df.get_naoaux()) == (g['nao'], g['naux'])
    assert abs(e - g[key]) < 1e-8, (e, g[key])


def test_config4_taxol_tensor_columns_vs_oracle():
    """BASELINE config 4 (taxol C47H51NO14 def2-TZVP, aux def2-tzvp-jkfit: nao 2228, naux 5598, nocc 226; geometry of
    data/taxol.xyz) as it is meant to run - aux rows sharded over 8 ranks: rank 3's rows of the tensor against oracle
    integrals on sampled AO shells (oracle (Q|pq), the oracle's own Cholesky factor of (P|Q) by trsm, rows [l0, l1)) to
    1e-9, and one J/K build of the shard (square image, 128-orbital chunks: nocc_pad = 240) against a dense FP64 product
    of the same rows."""
    import scipy.linalg
    import torch
    from pyscf_amd import gto, df
    from pyscf_amd.data import clusters
    from pyscf_amd.df import df_jk
    mol = gto.M(atom=clusters.taxol(), basis='def2-tzvp')
    obj = df.DF(mol)
    obj._shard_override = (3, 8)
    obj.build()
    nao, naux, nocc = mol.nao, obj.get_naoaux(), mol.nelectron // 2
    assert (mol.natm, nao, naux, nocc) == (113, 2228, 5598, 226)
    l0, l1 = obj.shard_range(naux, 3, 8)
    cd = obj._cderi_dev
    assert cd.shape == (l1 - l0, nao * (nao + 1) // 2) and l1 - l0 == 700
    aux = obj.auxmol
    low = scipy.linalg.cholesky(ref.int2c2e(aux), lower=True)
    loc = ref.ao_loc(mol)
    rng = np.random.RandomState(11)
    shells = sorted(set(rng.randint(0, mol.nbas, size=5).tolist()) | {0, mol.nbas - 1})
    worst, ncols = 0.0, 0
    for ish in shells:
        raw = ref.int3c2e_slab(mol, aux, ish, ish + 1)
        want = scipy.linalg.solve_triangular(low, raw, lower=True, overwrite_b=True, check_finite=False)[l0:l1]
        p0, p1 = int(loc[ish]), int(loc[ish + 1])
        pq0, pq1 = p0 * (p0 + 1) // 2, p1 * (p1 + 1) // 2
        worst = max(worst, float(np.abs(cd[:, pq0:pq1].cpu().numpy() - want).max()))
        ncols += pq1 - pq0
    assert ncols > 200 and worst < 1e-9, (ncols, worst)
    # J/K of the shard vs a dense product of the same rows
    dev = cd.device
    c = np.linalg.qr(np.random.default_rng(1).standard_normal((nao, nocc)))[0] * np.sqrt(2.0)
    dm = torch.from_numpy(c.dot(c.T)[None]).to(dev)
    vj, vk = df_jk.get_jk_device(obj, dm, [df_jk.pad_orbitals(c, dev)])
    assert obj._cderi_sq is not None
    idx = torch.tril_indices(nao, nao, device=dev)
    cdv = torch.from_numpy(c).to(dev)
    vk_ref = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    rho = torch.zeros(l1 - l0, dtype=torch.float64, device=dev)
    dmt = (dm[0] + dm[0].T)[idx[0], idx[1]]
    dmt[idx[0] == idx[1]] *= 0.5
    for b0 in range(0, l1 - l0, 100):
        full = torch.zeros((100, nao, nao), dtype=torch.float64, device=dev)
        full[:, idx[0], idx[1]] = cd[b0:b0 + 100]
        full = full + full.transpose(1, 2) - torch.diag_embed(torch.diagonal(full, dim1=1, dim2=2))
        x = torch.matmul(full, cdv)
        vk_ref += torch.einsum('Lpi,Lqi->pq', x, x)
        rho[b0:b0 + 100] = cd[b0:b0 + 100] @ dmt
    vj_ref = rho @ cd
    assert float((vk[0] - vk_ref).abs().max()) < 1e-11 * float(vk_ref.abs().max())
    assert float((vj[0] - vj_ref).abs().max()) < 1e-11 * float(vj_ref.abs().max())
    obj.reset()
    torch.cuda.empty_cache()
