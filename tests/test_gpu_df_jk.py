"""GPU parity: HIP J/K contraction (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from oracle import ref

pytestmark = pytest.mark.gpu


def _dfobj(mol, cderi, layout='packed'):
    """The tests of this module exercise the PACKED layout (the reference's rows + the optional square / diagonal-block images of
    r03, what a rank without 2x the tensor of HBM runs); the square layout has its own module (test_gpu_square_layout.py)."""
    from pyscf_amd import df
    obj = df.DF(mol)
    obj._cderi = cderi
    obj.layout = layout
    obj.build()
    return obj


@pytest.fixture(scope='module')
def h2o(h2o_dz):
    mol, aux = h2o_dz
    return mol, aux, ref.cholesky_eri(mol, aux)


def test_golden_jk_fingerprints(h2o):
    """pyscf/df/test/test_df_jk.py:144-152 (general-DM branch, hermi=0, two DMs)."""
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    np.random.seed(1)
    dms = np.random.random((2, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dms, hermi=0)
    assert abs(ref.fp(vj) - -194.15910890730066) < 1e-9
    assert abs(ref.fp(vk) - -46.365071587653517) < 1e-9
    vj0, vk0 = ref.get_jk(cderi, dms, hermi=0)
    assert np.abs(vj - vj0).max() < 1e-11
    assert np.abs(vk - vk0).max() < 1e-11
    vj1, _ = obj.get_jk(dms, hermi=0, with_k=False)
    _, vk1 = obj.get_jk(dms, hermi=0, with_j=False)
    assert np.abs(vj1 - vj0).max() < 1e-11 and np.abs(vk1 - vk0).max() < 1e-11


@pytest.mark.parametrize('k_square', ['auto', False])
def test_mo_branch(h2o, k_square):
    """k_square='auto': half transform on the unpacked image with the first J pass from its epilogue;
    False: packed-operand half transform and the two-pass J (what a rank without spare HBM runs)."""
    from pyscf_amd import lib
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    obj.k_square = k_square
    rng = np.random.default_rng(7)
    c = np.linalg.qr(rng.standard_normal((mol.nao, mol.nao)))[0]
    occ = np.zeros(mol.nao)
    occ[:5] = 2
    dm = (c * occ).dot(c.T)
    vj0, vk0 = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
    vj, vk = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
    assert np.abs(vj - vj0).max() < 1e-11
    assert np.abs(vk - vk0).max() < 1e-11
    assert (obj._cderi_sq is not None) == (k_square == 'auto')
    # untagged DM takes the general branch and must agree
    vj2, vk2 = obj.get_jk(dm, hermi=1)
    assert np.abs(vk2 - vk0).max() < 1e-11


@pytest.mark.parametrize('nao,naux,nocc', [(130, 301, 33), (257, 96, 161), (61, 17, 1)])
def test_random_tensor_ragged_sizes(nao, naux, nocc):
    """Ragged sizes (not multiples of the 128/16 tiles), nocc > 160 (two orbital chunks)."""
    from pyscf_amd import lib
    rng = np.random.default_rng(nao)
    npair = nao * (nao + 1) // 2
    cderi = rng.standard_normal((naux, npair)) / np.sqrt(nao)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c * occ).dot(c.T)
    obj = _dfobj(None, cderi)
    vj, vk = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
    full = ref.unpack_tril(cderi)
    rho = np.einsum('Lpq,pq->L', full, dm)
    vj0 = np.einsum('L,Lpq->pq', rho, full)
    tmp = full.dot(dm)                                     # [L][p][r]
    vk0 = np.einsum('Lpr,Lqr->pq', tmp, full, optimize=True)
    scale = max(1.0, np.abs(vk0).max())
    assert np.abs(vj - vj0).max() < 1e-10 * max(1.0, np.abs(vj0).max())
    assert np.abs(vk - vk0).max() < 1e-10 * scale
    dms = rng.standard_normal((3, nao, nao))
    vj, vk = obj.get_jk(dms, hermi=0)
    rho = np.einsum('Lpq,sqp->sL', full, dms)
    vj0 = np.einsum('sL,Lrt->srt', rho, full)
    f2 = full.reshape(-1, nao)
    vk0 = np.stack([(full.dot(d).transpose(0, 2, 1).reshape(-1, nao)).T.dot(f2) for d in dms])
    assert np.abs(vj - vj0).max() < 1e-10 * max(1.0, np.abs(vj0).max())
    assert np.abs(vk - vk0).max() < 1e-10 * max(1.0, np.abs(vk0).max())


def test_linearity_and_symmetry(h2o):
    """Size-independent properties: J,K linear in D; hermitian D -> hermitian J,K."""
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    rng = np.random.default_rng(3)
    a = rng.standard_normal((mol.nao, mol.nao)); a = a + a.T
    b = rng.standard_normal((mol.nao, mol.nao)); b = b + b.T
    ja, ka = obj.get_jk(a, 1)
    jb, kb = obj.get_jk(b, 1)
    jc, kc = obj.get_jk(2 * a - 3 * b, 1)
    assert np.abs(jc - (2 * ja - 3 * jb)).max() < 1e-10
    assert np.abs(kc - (2 * ka - 3 * kb)).max() < 1e-10
    assert np.abs(ja - ja.T).max() < 1e-12 and np.abs(ka - ka.T).max() < 1e-11


def test_golden_integral_direct_get_j(h2o):
    """J-only call before the tensor exists takes the integral-direct path and reproduces the same
    fingerprint (pyscf/df/test/test_df_jk.py:186-195; df_jk.get_j, df_jk.py:415-506)."""
    from pyscf_amd import df
    mol, aux, cderi = h2o
    obj = df.DF(mol, auxbasis='weigend')
    np.random.seed(1)
    dms = np.random.random((2, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dms, hermi=0, with_k=False)
    assert vk is None and obj._cderi_dev is None          # no tensor was built
    assert abs(ref.fp(vj) - -194.15910890730066) < 1e-9
    vj0, _ = ref.get_jk(cderi, dms, hermi=0, with_k=False)
    assert np.abs(vj - vj0).max() < 1e-11
    # several slabs give the same J
    obj2 = df.DF(mol, auxbasis='weigend')
    obj2._direct_slabs = lambda eng: [(i, i + 1) for i in range(eng.ao.n)]
    vj2, _ = obj2.get_jk(dms, hermi=0, with_k=False)
    assert np.abs(vj2 - vj0).max() < 1e-11


def test_edge_cases_empty_shard_zero_occ_and_eig_fallback(h2o):
    """Empty aux shard (more ranks than aux rows), a DM with no occupied orbitals, a single aux row, and
    the eigen-decomposition fallback for a linearly dependent fitting basis (df/incore.py:153-158,263-270)."""
    import torch
    from pyscf_amd import df, gto, lib
    from pyscf_amd.df import df_jk, incore
    mol, aux, cderi = h2o
    nao = mol.nao
    dev = torch.device('cuda', 0)
    # empty shard: contributes exact zeros
    obj = df.DF(mol)
    obj._cderi_dev = torch.zeros((0, cderi.shape[1]), dtype=torch.float64, device=dev)
    dm = np.eye(nao)
    vj, vk = obj.get_jk(dm, hermi=1)
    assert np.all(vj == 0) and np.all(vk == 0)
    # one aux row
    obj1 = df.DF(mol)
    obj1._cderi = cderi[:1]
    obj1.build()
    vj1, vk1 = obj1.get_jk(dm, hermi=1)
    vj0, vk0 = ref.get_jk(cderi[:1], dm, 1)
    assert np.abs(vj1 - vj0).max() < 1e-12 and np.abs(vk1 - vk0).max() < 1e-12
    # no occupied orbitals in the tagged DM -> K = 0, J from the (zero) DM = 0
    full = df.DF(mol)
    full._cderi = cderi
    full.build()
    z = lib.tag_array(np.zeros((nao, nao)), mo_coeff=np.eye(nao), mo_occ=np.zeros(nao))
    vjz, vkz = full.get_jk(z, hermi=1)
    assert np.all(vjz == 0) and np.all(vkz == 0)
    # linearly dependent aux basis (the same shells twice) -> Cholesky fails -> eig fallback, naux rows drop
    basis2 = {'O': gto.load_basis('weigend', 'O') * 2, 'H': gto.load_basis('weigend', 'H') * 2}
    aux2 = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis=basis2)
    cd2 = incore.cholesky_eri_gpu(mol, aux2, dev).cpu().numpy()
    assert cd2.shape[0] < aux2.nao                       # dependent functions were projected out
    assert np.abs(cd2.T.dot(cd2) - cderi.T.dot(cderi)).max() < 1e-6


def test_get_eri_ao2mo_and_npy_roundtrip(h2o, tmp_path):
    """pyscf/df/test/test_df.py:44-70 (test_ao2mo: four different MO blocks, then one square block) and
    :72-87 (_cderi_to_save / _cderi = file): DF.get_eri, DF.ao2mo and the saved tensor against plain
    numpy contractions of the oracle tensor."""
    from pyscf_amd import df, lib
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    nao = mol.nao
    eri4 = cderi.T.dot(cderi)
    assert np.abs(obj.get_eri() - lib.pack_tril(eri4)).max() < 1e-11
    b = lib.unpack_tril(cderi)                                   # (naux, nao, nao)
    np.random.seed(1)
    mos = np.random.random((nao, nao * 10))
    mos = (mos[:, :5], mos[:, 5:11], mos[:, 3:9], mos[:, 2:4])
    lij = np.einsum('pi,Lpq,qj->Lij', mos[0], b, mos[1]).reshape(len(b), -1)
    lkl = np.einsum('pi,Lpq,qj->Lij', mos[2], b, mos[3]).reshape(len(b), -1)
    got = obj.ao2mo(mos)
    assert got.shape == (30, 12) and np.abs(got - lij.T.dot(lkl)).max() < 1e-10
    mo = np.random.random((nao, nao))
    l1 = np.einsum('pi,Lpq,qj->Lij', mo, b, mo)
    ti, tj = np.tril_indices(nao)
    l1 = l1[:, ti, tj]
    got = obj.ao2mo(mo)
    assert got.shape == (nao * (nao + 1) // 2,) * 2 and np.abs(got - l1.T.dot(l1)).max() < 1e-9
    full = obj.ao2mo(mo, compact=False)
    assert full.shape == (nao * nao, nao * nao)
    # save / reload
    path = str(tmp_path / 'cderi.npy')
    obj.save(path)
    obj2 = df.DF(mol)
    obj2._cderi = path
    assert obj2.get_naoaux() == cderi.shape[0]
    assert np.abs(obj2.get_eri() - obj.get_eri()).max() < 1e-12
    blocks = np.vstack(list(obj2.loop(40)))
    assert np.abs(blocks - cderi).max() < 1e-14
    # HDF5 'j3c' (the reference's _cderi file, pyscf/df/df.py:97-99,185-199) through libhdf5
    from pyscf_amd.lib import hdf5
    if hdf5.available():
        h5 = str(tmp_path / 'cderi.h5')
        assert obj.save(h5) == h5 and hdf5.is_hdf5(h5)
        with hdf5.File(h5) as f:
            assert f['j3c'].shape == cderi.shape
        obj3 = df.DF(mol)
        obj3._cderi = h5
        assert obj3.get_naoaux() == cderi.shape[0]
        assert np.abs(np.vstack(list(obj3.loop(33))) - cderi).max() < 1e-14
        obj4 = df.DF(mol)
        obj4._cderi_to_save = str(tmp_path / 'auto.h5')           # written by build(), as the reference does
        obj4.build()
        assert hdf5.is_hdf5(obj4._cderi_to_save)
        # stock PySCF's default on-disk layout: GROUP 'j3c' with column blocks '0', '1', ... (outcore.cholesky_eri_b,
        # pyscf/df/outcore.py:215-221, read by df.py:227-241)
        import ctypes
        grp = str(tmp_path / 'blocks.h5')
        lib_h5 = hdf5._load()
        with hdf5.File(grp, 'w') as f:
            lib_h5.H5Gcreate2.restype = ctypes.c_int64
            g = lib_h5.H5Gcreate2(ctypes.c_int64(f._id), b'j3c', ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0))
            cuts = [0, 100, 101, cderi.shape[1]]
            for i in range(3):
                blk = np.ascontiguousarray(cderi[:, cuts[i]:cuts[i + 1]])
                f.create_dataset('j3c/%d' % i, blk.shape).write_rows(0, blk)
            lib_h5.H5Gclose(ctypes.c_int64(g))
        obj5 = df.DF(mol)
        obj5._cderi = grp
        assert np.abs(np.vstack(list(obj5.loop(33))) - cderi).max() < 1e-14
        from pyscf_amd import gto
        from tests.conftest import H2O
        bad = df.DF(gto.M(atom=H2O, basis='sto-3g'))             # wrong nao for this file: clear error, not an opaque IOError
        bad._cderi = grp
        with pytest.raises(RuntimeError, match='nao_pair'):
            bad.build()


def test_hermitian_dm_without_orbitals_is_factorized(h2o):
    """A symmetric DM without mo_coeff (initial guess, density differences) is split on the device into
    D = C+ C+^T - C- C-^T and sent through the MO kernels; must agree with the oracle's general formula and with the
    reference-style general branch (hermi=0), for a positive and for an indefinite matrix."""
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    rng = np.random.default_rng(3)
    a = rng.standard_normal((mol.nao, mol.nao))
    indefinite = a + a.T
    psd = a[:, :6].dot(a[:, :6].T)
    for dm in (psd, indefinite, np.array([psd, indefinite, -psd])):
        vj0, vk0 = ref.get_jk(cderi, dm, 1)
        vj1, vk1 = obj.get_jk(dm, hermi=1)
        vj2, vk2 = obj.get_jk(dm, hermi=0)
        scale = max(1.0, np.abs(vk0).max())
        assert np.abs(vj1 - vj0).max() < 1e-11 * scale and np.abs(vk1 - vk0).max() < 1e-11 * scale
        assert np.abs(vk2 - vk0).max() < 1e-11 * scale


def test_golden_uhf_veff_eight_density_matrices(h2o):
    """pyscf/df/test/test_df_jk.py:127-133: UHF veff of dm (2, 4, nao, nao), hermi=0: ||vhf|| = 413.82341595365853
    (V_s = J[D_a + D_b] - K[D_s], pyscf/scf/uhf.py:227-300) - eight DMs through the general branch in one call."""
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    np.random.seed(1)
    dm = np.random.random((2, 4, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dm.reshape(8, mol.nao, mol.nao), hermi=0)
    vj = vj.reshape(dm.shape)
    vk = vk.reshape(dm.shape)
    vhf = vj[0] + vj[1] - vk
    assert abs(np.linalg.norm(vhf) - 413.82341595365853) < 1e-8


def test_golden_assigned_cderi_from_exact_eri(h2o):
    """pyscf/df/test/test_df_jk.py:135-142: a tensor assigned by the caller (eigen-factorised exact ERIs, 300 'aux'
    rows) instead of built: DF-UHF then reproduces the exact energy -76.026765673110447."""
    import scipy.linalg
    from pyscf_amd import scf, df
    mol, aux, cderi = h2o
    nao = mol.nao
    eri = ref.int2e(mol)
    ti, tj = np.tril_indices(nao)
    eri4 = eri[ti, tj][:, ti, tj]
    w, u = scipy.linalg.eigh(eri4)
    idx = w > 1e-9
    mf = scf.UHF(mol).density_fit(auxbasis='weigend')
    mf.with_df._cderi = (u[:, idx] * np.sqrt(w[idx])).T.copy()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and abs(e - -76.026765673110447) < 1e-8, e


def test_lindep_metric_eig_fallback_reference_case():
    """pyscf/df/test/test_incore.py:171-179: an auxiliary basis with every H function listed twice makes the metric
    exactly singular; the eigen-decomposition fallback (lindep 1e-7, df/incore.py:153-158,263-270) must give the same
    fitted integrals cderi^T cderi as the non-redundant basis."""
    from pyscf_amd import gto, df
    mol = gto.M(atom=[('O', (0., 0., 0.)), ('H', (0., -0.757, 0.587)), ('H', (0., 0.757, 0.587))], basis='cc-pvdz')
    out = []
    for auxbasis in ('weigend', {'O': 'weigend', 'H': ('weigend', 'weigend')}):
        obj = df.DF(mol, auxbasis)
        obj.build()
        out.append(obj.get_eri())
    assert out[0].shape == out[1].shape
    assert np.abs(out[0] - out[1]).max() < 1e-9


def test_df_accepts_a_foreign_mole_object():
    """`pyscf_amd.df.DF(mol)` installed as `mf.with_df` of a stock PySCF mean-field object reads only the
    integral tables and a few plain attributes of the molecule - here a stand-in that has exactly those (the attribute list
    a real `pyscf.gto.Mole` also provides) and none of pyscf_amd.gto.Mole's methods."""
    import types
    from pyscf_amd import gto, df
    from tests.conftest import H2O
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    foreign = types.SimpleNamespace(_atm=mol._atm.copy(), _bas=mol._bas.copy(), _env=mol._env.copy(), _atom=list(mol._atom),
                                    atom=mol.atom, basis=mol.basis, charge=0, spin=0, max_memory=4000, stdout=None, verbose=0)
    obj = df.DF(foreign)
    rng = np.random.default_rng(2)
    dms = rng.standard_normal((2, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dms, hermi=0)                              # builds on first use, like the reference (df_jk.py:282-287)
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))
    vj0, vk0 = ref.get_jk(cderi, dms, 0)
    assert obj.get_naoaux() == 116 and np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9


@pytest.mark.parametrize('flags', [4, 8, 12])
@pytest.mark.parametrize('nao,naux,nocc', [(300, 64, 40), (700, 48, 160), (257, 96, 161)])
def test_syrk_variants_give_the_same_exchange(flags, nao, naux, nocc):
    """The optional SYRK schedules of K = X^T X - balanced k split (flag 4: full pieces + a short remainder piece per tile) and
    the re-tiled triangle without dead wave blocks (flag 8, csrc/df_jk.hip::syrk_slots_kernel; odd numbers of 64-column blocks:
    nao 300 -> 5, 700 -> 11, 257 -> 5 with a ragged last block) - against the default schedule and a plain numpy product."""
    from pyscf_amd import lib
    rng = np.random.default_rng(nao + flags)
    npair = nao * (nao + 1) // 2
    cderi = rng.standard_normal((naux, npair)) / np.sqrt(nao)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ)
    base = _dfobj(None, cderi)
    vj0, vk0 = base.get_jk(dm, hermi=1)
    obj = _dfobj(None, cderi)
    obj.k_syrk_flags = flags
    obj.k_block_bytes = 1 << 22                       # several K blocks: the partial buffers accumulate over launches
    vj, vk = obj.get_jk(dm, hermi=1)
    full = ref.unpack_tril(cderi)
    x = np.einsum('Lpq,qi->Lip', full, c[:, :nocc] * np.sqrt(2.0))
    vk_np = np.einsum('Lip,Liq->pq', x, x, optimize=True)
    scale = max(1.0, np.abs(vk_np).max())
    assert np.abs(vk - vk_np).max() < 1e-11 * scale and np.abs(vk - vk0).max() < 1e-11 * scale
    assert np.abs(vk - vk.T).max() == 0 and np.abs(vj - vj0).max() < 1e-11 * max(1.0, np.abs(vj0).max())


def test_partial_square_image():
    """k_square = 'auto' with room for only some rows (taxol on one GPU): the rows that have an unpacked image go through the
    square kernel, the rest through the packed-operand one, K blocks cut at the boundary; same J/K as without any image."""
    from pyscf_amd import lib
    nao, naux, nocc = 257, 301, 161
    rng = np.random.default_rng(5)
    cderi = rng.standard_normal((naux, nao * (nao + 1) // 2)) / np.sqrt(nao)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ)
    ref_obj = _dfobj(None, cderi)
    ref_obj.k_square = False
    vj0, vk0 = ref_obj.get_jk(dm, hermi=1)
    obj = _dfobj(None, cderi)
    obj.k_square_max_rows = 150                    # -> 128 rows (multiple of 64) get an image
    obj.k_block_bytes = 100 * 176 * 272 * 8        # ~100-row K blocks: 0-100, 100-128 | 128-228, 228-301
    vj, vk = obj.get_jk(dm, hermi=1)
    assert obj._cderi_sq is not None and obj._cderi_sq.shape[0] == 128
    # the packed rows beyond the image keep their diagonal blocks unpacked (DF.diag_image, r03); the reference object above ran
    # the same kernel on all rows, and once more without the side image
    assert obj._cderi_diag is not None and obj._diag_row0 == 128 and obj._cderi_diag.shape[0] == naux - 128
    assert ref_obj._cderi_diag is not None and ref_obj._diag_row0 == 0 and ref_obj._cderi_diag.shape[0] == naux
    # general-DM branch (hermi = 0) across the boundary: image rows are the second operand as they are, packed rows are unpacked
    dmg = rng.standard_normal((nao, nao))
    vjg, vkg = obj.get_jk(dmg, hermi=0)
    vjg0, vkg0 = ref.get_jk(cderi, dmg, 0)
    assert np.abs(vjg - vjg0).max() < 1e-11 * np.abs(vjg0).max() and np.abs(vkg - vkg0).max() < 1e-11 * np.abs(vkg0).max()
    plain = _dfobj(None, cderi)
    plain.k_square = plain.k_diag = False
    vj2, vk2 = plain.get_jk(dm, hermi=1)
    assert plain._cderi_diag is None
    assert np.abs(vj2 - vj0).max() < 1e-11 * max(1.0, np.abs(vj0).max()) and np.abs(vk2 - vk0).max() < 1e-11 * max(1.0, np.abs(vk0).max())
    assert np.abs(vj - vj0).max() < 1e-11 * max(1.0, np.abs(vj0).max()) and np.abs(vk - vk0).max() < 1e-11 * max(1.0, np.abs(vk0).max())


def test_foreign_tag_that_does_not_match_its_matrix(h2o):
    """Stock PySCF tags a density with mo_coeff / mo_occ and trusts the tag for K only (df_jk.py:339-381: J from the matrix,
    K from the orbitals, "#TODO: test whether dm.mo_coeff matching dm").  Here the first J pass is normally taken from the
    orbitals (fused into the half transform); a device-side probe that travels back with the results catches a tag that does
    not describe its matrix, and J is then redone from the matrix.  Consistent foreign tag: fused path, same numbers."""
    from pyscf_amd import lib
    mol, aux, cderi = h2o
    obj = _dfobj(mol, cderi)
    rng = np.random.default_rng(9)
    c = np.linalg.qr(rng.standard_normal((mol.nao, mol.nao)))[0]
    occ = np.zeros(mol.nao)
    occ[:5] = 2
    dm = (c * occ).dot(c.T)
    vj0, vk0 = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
    vj, vk = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)           # foreign but consistent
    assert obj._last_fused and np.abs(vj - vj0).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11
    other = dm + 0.05 * np.eye(mol.nao)                                                # the tag no longer matches the matrix
    vj1, _ = ref.get_jk(cderi, other, 1)
    vj, vk = obj.get_jk(lib.tag_array(other, mo_coeff=c, mo_occ=occ), hermi=1)
    assert np.abs(vj - vj1).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11
    # this package's OWN tag (make_rdm1: dm_from_orbitals) edited in place keeps its attributes: the probe - r05: on the
    # host, every 16th row for the own tag, beside the queued kernels, no upload of the matrix - still catches it
    own = lib.tag_array(dm.copy(), mo_coeff=c, mo_occ=occ, dm_from_orbitals=True)
    vj, vk = obj.get_jk(own, hermi=1)
    assert obj._last_fused and np.abs(vj - vj0).max() < 1e-11
    own *= 0.5
    assert getattr(own, 'dm_from_orbitals', False)
    vj, vk = obj.get_jk(own, hermi=1)
    assert np.abs(vj - 0.5 * vj0).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11


def test_second_j_pass_inside_the_syrk_kernel():
    """r05: PAMD_syrk_jfused - the second J pass of a K block folded INTO the re-tiled SYRK kernel of the same
    rows (syrk_slots_kernel<JF>: row-loads dealt to the workgroups by k-tile count, issued between the MFMA groups).  A shape with a
    fused form (odd number of 64-column blocks, <= 4 row-loads per k-tile), several K blocks, and a shape without one (falls back
    to the in-line pass): same J and K as the oracle."""
    from pyscf_amd import lib
    from pyscf_amd.df import df_jk
    for nao, naux, nocc, fused_form in ((300, 96, 120, True), (300, 40, 21, False), (200, 64, 90, False)):
        rng = np.random.default_rng(21)
        cderi = rng.standard_normal((naux, nao * (nao + 1) // 2)) / np.sqrt(nao)
        c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
        occ = np.zeros(nao)
        occ[:nocc] = 2
        dm = lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ)
        vj0, vk0 = ref.get_jk(cderi, np.asarray(dm), 1, mo_coeff=c, mo_occ=occ)
        for blocks in (1, 3):
            obj = _dfobj(None, cderi)
            obj.j2_policy = 'fused'
            if blocks > 1:
                obj.k_block_bytes = -(-naux // blocks) * ((nocc + 15) // 16 * 16) * ((nao + 15) // 16 * 16) * 8
            obj.kernel_timer = df_jk.KernelTimer()
            vj, vk = obj.get_jk(dm, hermi=1)
            names = set(obj.kernel_timer.summary())
            assert obj._last_fused
            assert ('vj_pass2' in names) == (not fused_form), (nao, nocc, names)      # the separate pass only where no fused form exists
            assert np.abs(vj - vj0).max() < 1e-11 * max(1.0, np.abs(vj0).max()), (nao, naux, nocc, blocks)
            assert np.abs(vk - vk0).max() < 1e-11 * max(1.0, np.abs(vk0).max()), (nao, naux, nocc, blocks)


def test_second_j_pass_schedules(h2o):
    """DF.j2_policy: the second J pass on the side stream beside a plain SYRK ('overlap'), in line before the re-tiled SYRK
    ('serial'), or whichever a one-off timing of both finds faster ('auto', tensors above j2_tune_min_bytes): same J and K."""
    from pyscf_amd import lib
    nao, naux, nocc = 200, 96, 37
    rng = np.random.default_rng(11)
    cderi = rng.standard_normal((naux, nao * (nao + 1) // 2)) / np.sqrt(nao)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ)
    vj0, vk0 = ref.get_jk(cderi, np.asarray(dm), 1, mo_coeff=c, mo_occ=occ)
    for policy in ('overlap', 'serial', 'auto', 'auto-lazy'):
        obj = _dfobj(None, cderi)
        obj.j2_policy = policy.split('-')[0]
        obj.j2_tune = 'lazy' if policy == 'auto-lazy' else 'eager'
        obj.j2_tune_min_bytes = 0                       # 'auto': time both schedules even on this small tensor
        obj.k_block_bytes = 40 * 48 * 208 * 8           # several K blocks
        # 'lazy' (the default, r06): the caller's own calls are the trials - priming call + 2 samples of each of the 3 candidates,
        # settled on the 8th call; 'eager': trial builds inside the first call
        for call in range(9 if policy == 'auto-lazy' else 2):
            vj, vk = obj.get_jk(dm, hermi=1)
            assert obj._last_fused
            assert np.abs(vj - vj0).max() < 1e-11 and np.abs(vk - vk0).max() < 1e-11, (policy, call)
            if policy == 'auto-lazy':
                assert hasattr(obj, '_j2_policy_times') == (call >= 7), call
        if policy.startswith('auto'):
            t = obj._j2_policy_times
            assert t['chosen'] in ('overlap', 'serial', 'fused') and t['overlap'] > 0 and t['serial'] > 0
            assert list(obj._j2_policy_cache.values()) == [t['chosen']]
            assert not getattr(obj, '_j2_lazy', {})
