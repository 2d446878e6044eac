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
    obj = df.DF(mol).build()
    torch.cuda.synchronize()
    return mol, obj


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
