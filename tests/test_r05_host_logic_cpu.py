"""CPU-only checks of the host logic added in round 5 (no GPU, no compute calls through the C ABI)."""
import gc
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pinned_pool_tracks_liveness_explicitly():
    """the result pools no longer ask sys.getrefcount.  A block is handed out again only when the array taken from it
    AND every view of that array are gone (weakref.finalize on the per-hand-out buffer object); safe under a lock."""
    from pyscf_amd.lib.pinned import PinnedPool
    keep = []

    def alloc(nbytes):
        a = np.zeros(nbytes // 8)
        keep.append(a)
        return a.ctypes.data, a
    pool = PinnedPool(alloc, max_idle=2)
    a, blk_a = pool.take(100)
    addr = a.ctypes.data
    view = a[10:20].reshape(5, 2)
    base_class_view = np.asarray(view).T
    del a
    gc.collect()
    b, _ = pool.take(100)
    assert b.ctypes.data != addr and blk_a.busy                  # a view is still alive: the block is NOT reused
    del view
    gc.collect()
    assert blk_a.busy
    del base_class_view
    gc.collect()
    assert not blk_a.busy
    c, _ = pool.take(90)                                         # fits the freed block (>= n, <= 4 n)
    assert c.ctypes.data == addr
    d, _ = pool.take(10)                                         # a much smaller request does not squat on a big block
    assert d.ctypes.data not in (addr, b.ctypes.data)
    assert pool.stats()['busy'] == 3
    # an allocator that refuses: None (callers fall back to pageable memory)
    assert PinnedPool(lambda n: (_ for _ in ()).throw(MemoryError())).take(5) is None


def test_mol_nao_for_any_molecule_like_object():
    import types
    from pyscf_amd import gto
    from pyscf_amd.df.df import _mol_nao
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='cc-pvdz')
    bare = types.SimpleNamespace(_atm=mol._atm, _bas=mol._bas, _env=mol._env)
    assert _mol_nao(mol) == _mol_nao(bare) == mol.nao == 24


def test_host_dm_probe_own_and_foreign_tags():
    """The tag probe of the host API (df_jk._host_dm_mismatch = the binding's probe): D = C C^T passes; an in-place edit of a tagged
    array is caught for the package's own tag exactly as for a foreign one - r06: the full matrix is probed for both
    (the every-16th-row probe of r05 missed a finite-difference edit dm[1, 2] += h; dm[2, 1] += h)."""
    from pyscf_amd.df import df_jk
    rng = np.random.default_rng(3)
    nao, nocc = 200, 37
    c = rng.standard_normal((nao, nocc))
    dm = c.dot(c.T)[None]
    assert df_jk._host_dm_mismatch(dm, [c], True) < 1e-12 and df_jk._host_dm_mismatch(dm, [c], False) < 1e-12
    assert df_jk._host_dm_mismatch(dm * 0.5, [c], True) > 1e-3
    edited = dm.copy()
    edited[0, 5, 7] += 1e-3                                      # one element, in a row the r05 sampled probe skipped
    assert df_jk._host_dm_mismatch(edited, [c], False) > 1e-8
    assert df_jk._host_dm_mismatch(edited, [c], True) > 1e-8     # own tag: seen as well
    fd = dm.copy()
    fd[0, 1, 2] += 1e-4
    fd[0, 2, 1] += 1e-4                                          # the advisor's finite-difference Fock example
    assert df_jk._host_dm_mismatch(fd, [c], True) > 1e-8
    h = df_jk._HostDM(dm, 'cpu')
    assert h.shape == dm.shape and h._t is None                  # nothing uploaded until a kernel asks for the matrix
    assert df_jk._dm_tensor(h).shape == dm.shape and h._t is not None


def test_bench_golden_density_is_the_generators():
    sys.path.insert(0, ROOT)
    import bench
    from oracle import golden_util
    dm, cfull, occ = bench._golden_density(120, 9)
    c = golden_util.synthetic_orbitals(120, 9) * np.sqrt(2.0)
    assert np.abs(dm - c.dot(c.T)).max() == 0 and np.abs(cfull[:, :9] - c).max() == 0 and occ.sum() == 9
    ri, ci = golden_util.sample_positions(120, 64)
    rng = np.random.RandomState(11)
    assert (rng.randint(0, 120, size=64) == ri).all() and (rng.randint(0, 120, size=64) == ci).all()

    class A:
        molecule, nwater, basis = 'water', 32, 'cc-pvtz'
    assert bench._golden_case(A, 1856, 160) == ('h2o32_ccpvtz_oracle.json', '', 160)
    A.nwater = 8                      # r06: the small case of the launch tests has its golden too
    assert bench._golden_case(A, 464, 40) == ('h2o8_ccpvtz_oracle.json', '', 40)
    A.nwater = 5
    assert bench._golden_case(A, 290, 25) is None
    out = bench._parity_golden(A, 290, 25, None)
    assert out['golden'] is None and 'why' in out


def test_cpu_baseline_calibration_picks_the_faster_blas():
    """oracle/ref_c.calibrate: both builds of the reference's C (scipy's OpenBLAS, libtorch_cpu's MKL) on a sample, the faster one
    is used; results of the two agree."""
    from oracle import ref_c
    if not (os.path.exists(ref_c.SO) and os.path.exists(ref_c.SO_MKL)):
        pytest.skip('oracle/_ref not built here')
    rng = np.random.RandomState(0)
    nao, naux, nocc = 48, 100, 9
    cd = rng.rand(naux, nao * (nao + 1) // 2) - .5
    c = np.linalg.qr(rng.rand(nao, nao))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c * occ).dot(c.T)
    old_cap, old_ok, old_var = ref_c.MAX_BLAS_CALLERS, ref_c._mkl_ok, ref_c._variant
    try:
        ref_c.MAX_BLAS_CALLERS = 1           # pretend the host has more cores than OpenBLAS admits callers: the MKL build is a candidate
        ref_c._mkl_ok = None
        cal = ref_c.calibrate(cd, dm, c, occ, rows=64)
        assert cal['chosen'] in ('openblas', 'mkl') and set(cal) >= {'openblas', 'chosen'}
        vj1, vk1, _ = ref_c.get_jk(cd, dm, c, occ, which='openblas')
        if 'mkl' in cal:
            vj2, vk2, _ = ref_c.get_jk(cd, dm, c, occ, which='mkl')
            assert np.abs(vj1 - vj2).max() < 1e-10 and np.abs(vk1 - vk2).max() < 1e-10
            assert cal['chosen'] == min(('openblas', 'mkl'), key=lambda k: cal[k]['ms_per_row'])
        assert 'calibration' in ref_c.describe() or len(cal) == 2
    finally:
        ref_c.MAX_BLAS_CALLERS, ref_c._mkl_ok, ref_c._variant = old_cap, old_ok, old_var


def test_hdf5_dataset_offset_allows_a_memmap():
    from pyscf_amd.lib import hdf5
    if not hdf5.available():
        pytest.skip('libhdf5 not found')
    import tempfile
    a = np.random.rand(9, 21)
    path = os.path.join(tempfile.mkdtemp(), 't.h5')
    with hdf5.File(path, 'w') as f:
        f.create_dataset('j3c', a.shape).write_rows(0, a)
    with hdf5.File(path) as f:
        off = f['j3c'].file_offset()
    assert off is not None
    mm = np.memmap(path, dtype='<f8', mode='r', offset=off, shape=a.shape)
    assert np.abs(mm - a).max() == 0 and mm[3:7].flags.c_contiguous


def test_native_from_rows_validates_its_arguments_without_a_device():
    from pyscf_amd import gto
    from pyscf_amd.df.native import NativeDF
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='sto-3g')
    with pytest.raises(ValueError):
        NativeDF.from_rows(mol, np.zeros((4, 5)))                # wrong nao_pair
    with pytest.raises(ValueError):
        NativeDF.from_rows(mol, np.zeros((4, 28), dtype=np.float32))
    obj = NativeDF(mol, shard=(3, 8))
    assert obj.shard == (3, 8) and obj._h is None
