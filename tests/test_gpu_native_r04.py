"""r04 additions to the host-array C ABI (include/pyscf_amd.h): PAMD_df_create_multi (aux index sharded over a device list in one
process), host-resident rows streamed under the kernels (out of core), omega tensors - driven from numpy in processes that never
import torch.  Reference interfaces: pyscf/df/df_jk.py:175-176,362-381 (the serial loop being sharded), pyscf/df/outcore.py:109-232,
pyscf/df/df.py:298-333."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _release_cached_hbm():
    """The workers are separate processes that hipMalloc for themselves: hand them what this process's torch allocator has cached."""
    if 'torch' in sys.modules:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    yield


def _run(script, marker, timeout, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', script)] + list(args), capture_output=True, text=True, timeout=timeout)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', script.replace('.py', '') + ('_' + args[0] if args else '') + '.log'), 'w') as f:
            f.write(p.stdout + p.stderr[-3000:])
    except OSError:
        pass
    assert p.returncode == 0 and marker in p.stdout, p.stdout[-3000:] + p.stderr[-4000:]
    return p.stdout


@pytest.mark.gpu
def test_multi_device_handle_streaming_and_omega_without_torch():
    _run('_native_abi_worker2.py', 'NATIVE_R04_OK', 1200)


@pytest.mark.gpu
def test_config3_through_the_native_handle_vs_oracle_golden():
    """(H2O)_32 cc-pVTZ J/K through PAMD_df_get_jk against tests/golden/h2o32_ccpvtz_oracle.json - the plain
    handle, two parts on the one test GPU, and with 40 % of the rows streamed from host memory."""
    _run('_native_cfg3_worker.py', 'NATIVE_CFG3_OK', 1500)


@pytest.mark.gpu
def test_xc_handle_grids_nr_rks_nr_uks_and_df_rks_golden_without_torch():
    """PAMD_grid_weights_host / PAMD_xc_create / PAMD_xc_nr_rks / PAMD_xc_nr_uks from numpy, no torch in the
    process; the reference's DF-RKS golden -76.690346887915879 (pyscf/dft/test/test_h2o.py:236-240) through NativeDF +
    NativeNumInt + NativeGrids."""
    _run('_native_xc_worker.py', 'NATIVE_XC_OK', 1200)


@pytest.mark.gpu
def test_stock_script_with_a_device_list_reaches_the_reference_golden():
    """`mf = scf.RHF(mol).density_fit(devices=[...]); mf.kernel()` - the documented single-process N > 1 recipe - and the
    PAMD_DEVICES environment form for an unmodified script: E = -76.025936299702536 (pyscf/df/test/test_df_jk.py:57-59)."""
    from pyscf_amd import gto, scf
    from pyscf_amd.df.native import NativeDF
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='cc-pvdz')
    mf = scf.RHF(mol).density_fit(auxbasis='weigend', devices=[0, 0, 0])
    assert isinstance(mf.with_df, NativeDF) and mf.with_df.devices == [0, 0, 0]
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and abs(e - -76.025936299702536) < 1e-8, e
    assert mf.with_df.layout()['parts'] == 3
    # Kohn-Sham: J/K and the XC tiles over the same device list (DF-RKS golden of pyscf/dft/test/test_h2o.py:236-240)
    from pyscf_amd import dft
    from pyscf_amd.dft import radi, gen_grid
    from pyscf_amd.dft.native import NativeNumInt
    old = radi.ATOM_SPECIFIC_TREUTLER_GRIDS
    radi.ATOM_SPECIFIC_TREUTLER_GRIDS = False
    try:
        m631 = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='6-31g')
        ks = dft.RKS(m631).density_fit(auxbasis='weigend', devices=[0, 0])
        assert isinstance(ks._numint, NativeNumInt) and ks._numint.devices == [0, 0]
        ks.grids.prune = gen_grid.treutler_prune
        ks.grids.atom_grid = {'H': (50, 194), 'O': (50, 194)}
        ks.xc = 'b88, vwn'
        ks.conv_tol = 1e-10
        eks = ks.kernel()
    finally:
        radi.ATOM_SPECIFIC_TREUTLER_GRIDS = old
    assert ks.converged and abs(eks - -76.690346887915879) < 1e-8, eks
    os.environ['PAMD_DEVICES'] = '0,0'
    try:
        m2 = scf.RHF(mol).density_fit(auxbasis='weigend')
    finally:
        del os.environ['PAMD_DEVICES']
    assert isinstance(m2.with_df, NativeDF) and m2.with_df.devices == [0, 0]
    m2.conv_tol = 1e-10
    assert abs(m2.kernel() - -76.025936299702536) < 1e-8


@pytest.mark.gpu
def test_config4_taxol_sharded_eight_ways_through_the_handle_vs_oracle_golden():
    """BASELINE config 4 as it is meant to run - aux-L sharded 8 ways - through the real multi-part code path (eight parts on the
    one test GPU: sharded build, eight host threads, gather, fixed-order sum) against the oracle-only taxol golden."""
    _run('_native_cfg45_worker.py', 'NATIVE_CONFIG4_OK', 1500, 'config4')


def test_library_exports_the_r04_handle_api_without_torch():
    code = ("import sys; sys.path.insert(0, %r); from pyscf_amd.df import native; lib = native.load(); "
            "[getattr(lib, n) for n in ('PAMD_df_create_ex', 'PAMD_df_create_multi', 'PAMD_df_layout', 'PAMD_grid_weights_host', 'PAMD_xc_create', 'PAMD_xc_nr_rks', 'PAMD_xc_nr_uks', 'PAMD_xc_plan_info', 'PAMD_xc_destroy')]; from pyscf_amd.dft import native as xn; "
            "assert 'torch' not in sys.modules; print('ok')" % ROOT)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'ok' in p.stdout, p.stderr[-2000:]


@pytest.mark.gpu
def test_torch_resident_df_falls_back_to_the_streaming_handle_when_the_tensor_does_not_fit():
    """The out-of-core twin behind the production object (pyscf/df/outcore.py:109-232, df/df.py:167): a DF whose tensor is above
    its device-memory cap no longer raises MemoryError - build() hands it to the C handle (rows in HBM + page-locked host rows
    streamed per build); J/K and the SCF energy are those of the in-core object, the device SCF loop steps aside."""
    import numpy as np
    from oracle import ref
    from pyscf_amd import gto, scf, df
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(2), basis='cc-pvtz')
    nao = mol.nao
    npair = nao * (nao + 1) // 2
    obj = df.DF(mol)
    obj.outcore_device_bytes = 40 * npair * 8 * 4            # ~40 resident rows of 278
    obj.build()
    lay = obj.out_of_core()
    assert lay is not None and lay['rows_host'] > 0 and lay['rows_resident'] + lay['rows_host'] == obj.get_naoaux() == 278, lay
    cd = ref.cholesky_eri(mol, df.make_auxmol(mol))
    rng = np.random.RandomState(2)
    dm = rng.rand(2, nao, nao)
    vj0, vk0 = ref.get_jk(cd, dm, 0)
    vj, vk = obj.get_jk(dm, hermi=0)
    assert np.abs(vj - vj0).max() < 1e-10 and np.abs(vk - vk0).max() < 1e-10
    assert np.abs(np.vstack(list(obj.loop(64))) - cd).max() < 1e-10
    # r06 (VERDICT r05 Missing 4): the tensor's consumers work from loop() blocks, as the reference's do (pyscf/df/df.py:269-296) -
    # get_eri and ao2mo on the out-of-core object equal the in-core ones
    inc = df.DF(mol).build()
    co = np.linalg.qr(rng.standard_normal((nao, 7)))[0]
    cv = np.linalg.qr(rng.standard_normal((nao, 5)))[0]
    assert np.abs(obj.ao2mo((co, cv, co, cv)) - inc.ao2mo((co, cv, co, cv))).max() < 1e-10
    assert np.abs(obj.ao2mo(co) - inc.ao2mo(co)).max() < 1e-10
    small = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='cc-pvdz')
    o1 = df.DF(small)
    o1.outcore_device_bytes = 30 * 300 * 8
    o1.build()
    assert o1.out_of_core() is not None and o1.out_of_core()['rows_host'] > 0
    assert np.abs(o1.get_eri() - df.DF(small).build().get_eri()).max() < 1e-10
    del inc, o1
    mf = scf.RHF(mol).density_fit(with_df=obj)
    mf.device_scf_min_nao = 0                                # would take the HBM-resident loop; must fall back to the host loop
    mf.conv_tol = 1e-10
    e = mf.kernel()
    m0 = scf.RHF(mol).density_fit()
    m0.conv_tol = 1e-10
    e0 = m0.kernel()
    assert mf.converged and abs(e - e0) < 1e-9, (e, e0)
    with pytest.raises(NotImplementedError):
        mf.nuc_grad_method().kernel()
    obj.reset()
    assert obj.out_of_core() is None
    # switched off: the old behaviour
    obj2 = df.DF(mol)
    obj2.outcore_device_bytes = 40 * npair * 8 * 4
    obj2.outcore = False
    with pytest.raises(MemoryError):
        obj2.build()


@pytest.mark.gpu
def test_cderi_file_is_streamed_from_an_mmap_when_it_does_not_fit(tmp_path):
    """r05: `DF._cderi = 'file.h5'` (PySCF's own format: dataset 'j3c' (naux, nao_pair), pyscf/df/df.py:97-99,
    outcore.py:217-221) whose rows do not fit the device is NOT loaded: the dataset is mapped and the C handle streams the
    non-resident rows out of the mapping in every build (the reference reads its file block by block in DF.loop, df.py:214-242).
    Also `_cderi = ndarray` and `_cderi_to_save` on the out-of-core path."""
    import numpy as np
    from oracle import ref
    from pyscf_amd import gto, df
    from pyscf_amd.data import clusters
    from pyscf_amd.lib import hdf5
    if not hdf5.available():
        pytest.skip('libhdf5 not found')
    mol = gto.M(atom=clusters.water_cluster(2), basis='cc-pvtz')
    nao = mol.nao
    npair = nao * (nao + 1) // 2
    cd = ref.cholesky_eri(mol, df.make_auxmol(mol))
    naux = cd.shape[0]
    path = str(tmp_path / 'cderi.h5')
    with hdf5.File(path, 'w') as f:
        f.create_dataset('j3c', (naux, npair)).write_rows(0, cd)
    rng = np.random.RandomState(5)
    dm = rng.rand(2, nao, nao)
    vj0, vk0 = ref.get_jk(cd, dm, 0)
    cap = 60 * npair * 8 * 4
    for src in (path, cd):
        obj = df.DF(mol)
        obj._cderi = src
        obj.outcore_device_bytes = cap
        obj.build()
        lay = obj.out_of_core()
        assert lay is not None and obj._cderi_dev is None and lay['rows_host'] > 0 and lay['rows_resident'] + lay['rows_host'] == naux, lay
        assert obj.get_naoaux() == naux
        vj, vk = obj.get_jk(dm, hermi=0)
        assert np.abs(vj - vj0).max() < 1e-10 and np.abs(vk - vk0).max() < 1e-10
        assert np.abs(np.vstack(list(obj.loop(50))) - cd).max() == 0
        obj.reset()
    # the in-core route of the same file is unchanged
    obj = df.DF(mol)
    obj._cderi = path
    obj.build()
    assert obj.out_of_core() is None and obj._cderi_dev.shape == (naux, npair)
    obj.reset()
    # `_cderi_to_save` on the out-of-core BUILD path: the handle's row blocks go to the file
    out = str(tmp_path / 'saved.h5')
    obj = df.DF(mol)
    obj.outcore_device_bytes = cap
    obj._cderi_to_save = out
    obj.build()
    assert obj.out_of_core() is not None
    with hdf5.File(out) as f:
        back = f['j3c'].read_rows(0, naux)
    assert np.abs(back - cd).max() < 1e-10
    obj.reset()
