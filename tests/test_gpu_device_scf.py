"""The HBM-resident SCF loop (pyscf_amd/scf/device_scf.py) against the host loop that mirrors pyscf/scf/hf.py:49-241 line by
line, and against the oracle's SCF: same converged energies, orbitals that diagonalise the final Fock matrix, SP2 purification
really in use."""
import numpy as np
import pytest

from oracle import ref


def test_sp2_purification_equals_eigenprojector_cpu():
    """purify_sp2 on the CPU (torch): projector on the nocc lowest eigenvectors, several spectra / fillings."""
    import torch
    from pyscf_amd.scf.device_scf import purify_sp2
    rng = np.random.default_rng(5)
    for n, nocc, gap in ((40, 7, 0.4), (64, 32, 0.05), (50, 1, 1.0), (30, 29, 0.3)):
        q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        w = np.sort(rng.uniform(-20, 30, n))
        w[nocc:] += gap - (w[nocc] - w[nocc - 1])                      # set the gap
        a = (q * w).dot(q.T)
        p, it = purify_sp2(torch.from_numpy(a), nocc)
        assert p is not None, (n, nocc)
        p0 = q[:, :nocc].dot(q[:, :nocc].T)
        assert np.abs(p.numpy() - p0).max() < 1e-9, (n, nocc, it, np.abs(p.numpy() - p0).max())
    # no gap at the Fermi level: must report failure, not a wrong projector
    w = np.arange(10.0)
    w[4] = w[3]
    a = np.diag(w)
    p, it = purify_sp2(torch.from_numpy(a), 4, max_iter=60)
    assert p is None


def _run(mol, xc, device, **kw):
    from pyscf_amd import scf, dft
    mf = (dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)).density_fit()
    if xc:
        mf.grids.level = 1
    mf.conv_tol = 1e-10
    mf.device_scf = device
    mf.device_scf_min_nao = 0
    for k, v in kw.items():
        setattr(mf, k, v)
    e = mf.kernel()
    return mf, e


@pytest.mark.gpu
@pytest.mark.parametrize('xc', ['', 'b3lyp', 'pbe', 'lda,vwn'])
def test_device_loop_equals_host_loop(xc):
    from pyscf_amd import gto
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')
    mh, eh = _run(mol, xc, False)
    md, ed = _run(mol, xc, True)
    assert mh.converged and md.converged
    assert abs(eh - ed) < 2e-9, (xc, eh, ed)
    assert getattr(md, '_purify_iters', 0) > 0                          # the purification path was taken
    # full eigh everywhere (purify off) gives the same again
    me, ee = _run(mol, xc, True, purify=False)
    assert abs(ee - eh) < 2e-9
    # mo_energy / mo_coeff at the API edge: numpy, orthonormal, and they diagonalise the final Fock matrix
    c, e = md.mo_coeff, md.mo_energy
    assert isinstance(c, np.ndarray) and isinstance(e, np.ndarray) and isinstance(md.mo_occ, np.ndarray)
    s = md.get_ovlp()
    assert np.abs(c.T.dot(s).dot(c) - np.eye(c.shape[1])).max() < 1e-9
    assert md.mo_occ.sum() == mol.nelectron and np.all(np.diff(e) > -1e-9)
    f = md.get_hcore() + md.get_veff(mol, md.make_rdm1())
    r = f.dot(c) - s.dot(c) * e
    assert np.abs(r[:, md.mo_occ > 0]).max() < 5e-5                      # converged to conv_tol_grad = 1e-5
    assert np.abs(e - mh.mo_energy).max() < 1e-5


@pytest.mark.gpu
def test_device_loop_vs_oracle_scf():
    """DF-RHF of (H2O)_2 cc-pVDZ: the oracle's own SCF (oracle/ref.rhf_kernel, core guess) and the device loop (minao guess)."""
    from pyscf_amd import gto, df
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(2), basis='cc-pvdz')
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))
    conv, e0 = ref.rhf_kernel(mol, lambda d, c, o: (lambda v: v[0] - .5 * v[1])(ref.get_jk(cderi, d, 1)), conv_tol=1e-11)[:2]
    assert conv
    md, ed = _run(mol, '', True)
    assert abs(ed - e0) < 1e-8, (ed, e0)


@pytest.mark.gpu
def test_device_loop_sp2_on_the_syrk_kernel():
    """From 256 orbitals the purification squares its matrix on the product's own SYRK kernel (device_scf._SymSquare) instead of a
    library GEMM: (H2O)_5 cc-pVTZ, 290 orbitals zero-padded to 304, device loop = host loop."""
    from pyscf_amd import gto
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(5), basis='cc-pvtz')
    assert mol.nao == 290
    mh, eh = _run(mol, '', False)
    md, ed = _run(mol, '', True)
    assert mh.converged and md.converged and abs(eh - ed) < 2e-9, (eh, ed)
    assert getattr(md, '_purify_iters', 0) > 0
    # the squaring itself
    import torch
    from pyscf_amd.scf.device_scf import _SymSquare
    a = torch.randn(290, 290, dtype=torch.float64, device='cuda')
    a = (a + a.T) * 0.5
    sq = _SymSquare(290, a.device)
    x2 = sq.square(sq.pad(a, 0), 1)
    assert float((x2[:290, :290] - a @ a).abs().max()) < 1e-11 * float((a @ a).abs().max())
    assert float(x2[290:].abs().max()) == 0 and float((x2 - x2.T).abs().max()) == 0


@pytest.mark.gpu
def test_unconverged_device_loop_returns_the_orbitals_of_its_density():
    """stopped at max_cycle (or with conv_check = False) the loop returns, like hf.py:176-241, the orbitals e_tot and
    dm were made from - not the eigenvectors of the next Fock matrix.  Same numbers as the host loop cycle for cycle."""
    from pyscf_amd import gto
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')
    for kw in (dict(max_cycle=4), dict(conv_check=False)):
        mh, eh = _run(mol, '', False, **kw)
        md, ed = _run(mol, '', True, **kw)
        assert md.converged == mh.converged
        assert abs(eh - ed) < 1e-8, (kw, eh, ed)
        dm = md.make_rdm1()
        assert abs(md.energy_tot(dm) - ed) < 1e-9, kw                   # the returned orbitals reproduce the returned energy
        assert np.abs(md.mo_energy - mh.mo_energy).max() < 1e-6, kw
        assert np.abs(np.asarray(dm) - np.asarray(mh.make_rdm1())).max() < 1e-6, kw


@pytest.mark.gpu
def test_overridden_methods_keep_the_host_loop():
    """an instance- or class-level override of a method the device loop restates (the PySCF idiom mf.get_occ = ...)
    must not be bypassed."""
    from pyscf_amd import gto, scf
    from pyscf_amd.data import clusters
    from pyscf_amd.scf import device_scf
    mol = gto.M(atom=clusters.water_cluster(2), basis='cc-pvdz')
    mf = scf.RHF(mol).density_fit()
    mf.device_scf_min_nao = 0
    assert device_scf.eligible(mf)
    calls = []
    stock = mf.get_occ

    def get_occ(mo_energy=None, mo_coeff=None):
        calls.append(1)
        return stock(mo_energy, mo_coeff)
    mf.get_occ = get_occ
    assert not device_scf.eligible(mf)
    mf.kernel()
    assert mf.converged and len(calls) >= mf.cycles

    class MyRHF(scf.RHF):
        def get_fock(self, *a, **k):
            return scf.RHF.get_fock(self, *a, **k)
    MyRHF.__name__ = 'RHF'
    m2 = MyRHF(mol).density_fit()
    m2.device_scf_min_nao = 0
    assert not device_scf.eligible(m2)


@pytest.mark.gpu
@pytest.mark.parametrize('xc', ['', 'b3lyp', 'pbe'])
def test_device_loop_over_the_c_handle(xc):
    """r06 (VERDICT r05 item 6 / Missing 4): the HBM-resident loop with `with_df` = the host-array C handle - PAMD_df_get_jk with
    DEVICE pointers (flags bit 3) through NativeDF.get_jk_device.  Same energies as the loop over df.DF and as the host loop over
    the same handle; J and K of the device-pointer call equal the host-array call's (pyscf/df/df_jk.py:280-413)."""
    import torch
    from pyscf_amd import gto, scf, dft, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df.native import NativeDF
    from pyscf_amd.scf import device_scf
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')

    def run(device):
        mf = (dft.RKS(mol, xc=xc) if xc else scf.RHF(mol))
        mf = mf.density_fit(with_df=NativeDF(mol))
        if xc:
            mf.grids.level = 1
        mf.conv_tol = 1e-10
        mf.device_scf = device
        mf.device_scf_min_nao = 0
        assert device_scf.eligible(mf) == device
        return mf, mf.kernel()
    md, ed = run(True)
    mh, eh = run(False)
    _, e0 = _run(mol, xc, True)
    assert md.converged and mh.converged
    assert abs(ed - eh) < 2e-9 and abs(ed - e0) < 2e-9, (xc, ed, eh, e0)
    assert getattr(md, '_purify_iters', 0) > 0 and getattr(md, '_dm_dev', None) is not None      # the device loop really ran
    # the device-pointer call against the host-array call of the same handle
    h = md.with_df
    occ = md.mo_occ > 0
    orbo = np.ascontiguousarray(md.mo_coeff[:, occ] * np.sqrt(md.mo_occ[occ]))
    dm = orbo.dot(orbo.T)
    vj0, vk0 = h.get_jk(lib.tag_array(dm, mo_coeff=md.mo_coeff, mo_occ=md.mo_occ), hermi=1)
    dev = torch.device('cuda', h.device_index())
    vj, vk = h.get_jk_device(torch.from_numpy(dm).to(dev), torch.from_numpy(orbo).to(dev))
    assert np.abs(vj.cpu().numpy() - vj0).max() < 1e-11 and np.abs(vk.cpu().numpy() - vk0).max() < 1e-11
    vj1, vk1 = h.get_jk_device(torch.from_numpy(dm).to(dev), None, with_k=False)                 # J alone: from the matrix
    assert vk1 is None and np.abs(vj1.cpu().numpy() - vj0).max() < 1e-11
