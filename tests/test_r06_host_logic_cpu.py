"""Host-side logic added in r06 that needs no GPU: the collective out-of-core / layout decision (gloo, world 2), the HDF5 datatype
probe, the clear errors of a from_rows handle, the re-entrant pinned pool, the bounded matrix-vector probe, the HBM book-keeping,
the XCD-aware SYRK order table, the square-layout bookkeeping of df.DF."""
import ctypes
import gc
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _agree_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyscf_amd import df
    obj = df.DF(None)
    # rank 1 says "does not fit": MIN over the ranks -> nobody stays in core; both say "fits" -> everybody does
    a = obj._all_ranks_agree(rank == 0)
    b = obj._all_ranks_agree(True)
    c = obj._all_ranks_agree(False)
    q.put((rank, a, b, c))
    dist.destroy_process_group()


def test_out_of_core_decision_is_collective_over_the_ranks():
    """each rank used to decide from its own memory query; ranks that disagreed issued different collectives."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, False, True, False), (1, False, True, False)]
    # without a process group the flag is the rank's own
    from pyscf_amd import df
    assert df.DF(None)._all_ranks_agree(True) is True and df.DF(None)._all_ranks_agree(False) is False


def test_hdf5_datatype_probe_refuses_anything_but_little_endian_float64(tmp_path):
    from pyscf_amd.lib import hdf5
    if not hdf5.available():
        pytest.skip('no libhdf5 in this image')
    path = str(tmp_path / 't.h5')
    with hdf5.File(path, 'w') as f:
        f.create_dataset('j3c', (3, 6)).write_rows(0, np.arange(18.).reshape(3, 6))
    with hdf5.File(path) as f:
        assert f['j3c'].is_native_f64_le() and f['j3c'].file_offset() is not None
    # a float32 dataset written through the C API directly
    lib = hdf5._load()
    hid = hdf5._hid
    f32 = hid.in_dll(lib, 'H5T_NATIVE_FLOAT_g').value
    p2 = str(tmp_path / 'f32.h5')
    fid = lib.H5Fcreate(p2.encode(), 2, hid(0), hid(0))            # H5F_ACC_TRUNC
    assert fid >= 0
    dims = (ctypes.c_uint64 * 2)(3, 6)
    sp = lib.H5Screate_simple(2, dims, None)
    ds = lib.H5Dcreate2(hid(fid), b'j3c', hid(f32), hid(sp), hid(0), hid(0), hid(0))
    assert ds >= 0
    data = np.arange(18, dtype=np.float32)
    assert lib.H5Dwrite(hid(ds), hid(f32), hid(0), hid(0), hid(0), data.ctypes.data_as(ctypes.c_void_p)) >= 0
    lib.H5Dclose(hid(ds))
    lib.H5Sclose(hid(sp))
    lib.H5Fclose(hid(fid))
    with hdf5.File(p2) as f:
        assert not f['j3c'].is_native_f64_le()


def test_from_rows_handle_refuses_range_coulomb_and_reset_mol():
    from pyscf_amd.df.native import NativeDF
    obj = NativeDF.__new__(NativeDF)
    obj.__init__(None)
    obj.auxmol = False                       # what from_rows leaves: no auxiliary molecule, the rows came ready-made
    with pytest.raises(NotImplementedError, match='ready-made'):
        obj.range_coulomb(0.3)
    with pytest.raises(NotImplementedError, match='ready-made'):
        obj.reset(mol=object())
    assert obj.range_coulomb(0) is obj and obj.reset() is obj


def test_pinned_pool_release_is_reentrant_under_its_own_lock():
    """the finalizer of a result array may fire on the thread that is inside take() (cyclic GC): the lock is re-entrant."""
    from pyscf_amd.lib.pinned import PinnedPool
    bufs = []

    def alloc(nbytes):
        b = (ctypes.c_char * nbytes)()
        bufs.append(b)
        return ctypes.addressof(b), b
    pool = PinnedPool(alloc)
    arr, blk = pool.take(16)
    with pool._lock:                          # as if a GC run inside take() collected an earlier result
        pool._release(blk)
    assert blk.busy is False
    del arr
    gc.collect()
    a2, b2 = pool.take(16)
    assert b2 is blk and pool.stats() == {'blocks': 1, 'busy': 1}


def test_bounded_matvec_and_full_probe():
    from pyscf_amd import lib
    rng = np.random.default_rng(0)
    m, v = rng.standard_normal((300, 300)), rng.standard_normal(300)
    assert np.allclose(lib.bounded_matvec(m, v), m.dot(v), rtol=0, atol=1e-12)
    c = rng.standard_normal((300, 40))
    dm = c.dot(c.T)[None]
    assert lib.dm_orbital_mismatch(dm, [c]) < 1e-12
    fd = dm.copy()
    fd[0, 1, 2] += 1e-4
    fd[0, 2, 1] += 1e-4
    assert lib.dm_orbital_mismatch(fd, [c]) > 1e-8


def test_hbm_bookkeeping():
    from pyscf_amd.lib import hbm
    hbm.hold(0, 'xc_image', 123)
    hbm.hold(torch.device('cuda', 1) if False else 1, 'xc_image', 7)
    assert hbm.held(0, 'xc_image') == 123 and hbm.held(1, 'xc_image') == 7 and hbm.held(0, 'other') == 0
    hbm.drop(0, 'xc_image')
    hbm.drop(1, 'xc_image')
    assert hbm.held(0, 'xc_image') == 0


def test_square_layout_bookkeeping_without_a_device():
    """df.DF: `_cderi_dev` is a property over the packed rows; has_tensor / tensor_shape never materialise anything; the reserve of
    the one HBM budget subtracts what the XC plan already holds; estimate of the compact AO image."""
    from pyscf_amd import df, gto
    from pyscf_amd.dft.numint import estimate_ao_image_bytes
    from pyscf_amd.lib import hbm
    obj = df.DF(None)
    assert not obj.has_tensor() and obj._cderi_dev is None and obj._layout is None
    t = torch.zeros((5, 6), dtype=torch.float64)
    obj._cderi_dev = t
    assert obj.has_tensor() and obj._layout == 'packed' and obj.tensor_shape() == (5, 6) and obj._cderi_dev is t
    assert obj.packed_rows(1, 3).shape == (2, 6)
    obj.reset()
    assert not obj.has_tensor() and obj._layout is None
    df.DF.SQ_STRIDE_PAD, keep = 32, df.DF.SQ_STRIDE_PAD
    sq = df.DF.alloc_square(3, 16, 'cpu')
    df.DF.SQ_STRIDE_PAD = keep
    assert sq.shape == (3, 16, 16) and sq.stride() == (16 * 16 + 32, 16, 1) and float(sq.abs().sum()) == 0
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='cc-pvdz')
    est = estimate_ao_image_bytes(mol)
    assert est == 12500 * 3 * 4 * 8 * 24
    o2 = df.DF(mol)
    o2.device = 'cpu'
    o2.k_block_bytes = 1 << 20
    base = o2._reserve_after_build(10, 32)
    o2.xc_image_hint = 1000
    assert o2._reserve_after_build(10, 32) == base + 1000 + (12 << 30)
    hbm.hold('cpu', 'xc_image', 400)
    try:
        assert o2._reserve_after_build(10, 32) == base + 600 + (4 << 30)        # the plan exists: its work buffers are allocated already
    finally:
        hbm.drop('cpu', 'xc_image')


def test_kohn_sham_density_fit_tells_the_tensor_about_the_xc_image():
    from pyscf_amd import gto, scf, dft
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='sto-3g')
    assert scf.RHF(mol).density_fit().with_df.xc_image_hint == 0
    ks = dft.RKS(mol, xc='b3lyp').density_fit()
    assert ks.with_df.xc_image_hint > 0
    ku = dft.UKS(mol, xc='b3lyp').density_fit(devices=[0])
    assert ku.with_df.xc_image_hint > 0          # the host-array handle object carries it into PAMD_df_options.reserve_bytes


def test_grid_box_sort_is_the_reference_permutation():
    """gen_grid.arg_group_grids: one stable sort of a scalar key = the reference's numpy.unique(box_ids, axis=0) ranks + stable argsort
    (pyscf/dft/gen_grid.py:369-388), point for point."""
    from pyscf_amd import gto
    from pyscf_amd.dft import gen_grid
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587; O 4.1 0.3 -2.2; H 4.1 1.1 -1.7; H 3.4 0.4 -2.8', basis='sto-3g')
    rng = np.random.default_rng(3)
    coords = rng.uniform(-14.0, 14.0, size=(20000, 3))             # some far outside the padded molecular box: clipped box ids
    got = gen_grid.arg_group_grids(mol, coords)
    ac = mol.atom_coords()
    lo, hi = ac.min(axis=0) - gen_grid.GROUP_BOUNDARY_PENALTY, ac.max(axis=0) + gen_grid.GROUP_BOUNDARY_PENALTY
    boxes = ((hi - lo) * (1. / gen_grid.GROUP_BOX_SIZE)).round().astype(int)
    box_ids = np.floor((coords - lo) * (1. / ((hi - lo) / boxes))).astype(int)
    box_ids[box_ids < -1] = -1
    for k in range(3):
        box_ids[box_ids[:, k] > boxes[k], k] = boxes[k]
    want = np.unique(box_ids, axis=0, return_inverse=True)[1].ravel().argsort(kind='stable')
    assert np.array_equal(got, want)


def test_sparse_plan_tables_vectorised_equal_the_per_tile_loop():
    """dft/sparse_grid._tables: ld / idx / nsub from one boolean [tile][function] table = the per-tile concatenation of the active
    shells' function ranges (padding columns = nao, ld rounded up to 16, an empty tile keeps 16 columns)."""
    from pyscf_amd.dft import sparse_grid
    rng = np.random.default_rng(5)
    nsh, nloc = 37, 61
    l = rng.integers(0, 4, size=nsh)
    nfn = 2 * l + 1
    ao0 = np.concatenate([[0], np.cumsum(nfn)[:-1]])
    nao = int(nfn.sum())
    active = rng.random((nloc, nsh)) < 0.35
    active[7] = False
    plan = sparse_grid.SparsePlan.__new__(sparse_grid.SparsePlan)
    plan.nloc, plan.nao, plan.G, plan.dev = nloc, nao, 512, torch.device('cpu')
    plan._tables({'ao0': torch.from_numpy(ao0), 'l': torch.from_numpy(l)}, active)
    ld, rows = [], []
    for i in range(nloc):
        fns = np.concatenate([np.arange(ao0[s], ao0[s] + nfn[s]) for s in np.nonzero(active[i])[0]] + [np.zeros(0, int)])
        n = (max(len(fns), 1) + 15) // 16 * 16
        row = np.full(n, nao, np.int32)
        row[:len(fns)] = fns
        ld.append(n)
        rows.append(row)
    assert np.array_equal(plan.ld_host, np.array(ld)) and np.array_equal(plan.idx.numpy(), np.concatenate(rows))
    assert np.array_equal(plan.nsub_host, np.array([(r < nao).sum() for r in rows]))
    assert np.array_equal(plan.idx_off_host, np.concatenate([[0], np.cumsum(ld)[:-1]]))


def test_library_rules_shared_by_both_host_layers():
    """r06 (first steps): the K-block size, the SYRK plan and the J2 schedule decision are library functions
    that df_jk (torch layer) and df_handle.hip (C handle) both call - here against the formulas they replaced."""
    import itertools
    from pyscf_amd import lib
    so = lib.load_library()
    so.PAMD_k_block_rows.restype = ctypes.c_long

    def old_blk(naux, rows, ldx, budget):
        blk = max(1, int(budget // (rows * ldx * 8)))
        blk = min(blk, max(naux, 1))
        nblk = -(-max(naux, 1) // blk)
        return -(-max(naux, 1) // nblk)
    for naux, rows, ldx, b in itertools.product((0, 1, 17, 556, 4448, 5598, 14848), (16, 160, 240, 640), (64, 1856, 2240, 3072),
                                                (1 << 20, 4 << 30, 12 << 30)):
        got = so.PAMD_k_block_rows(ctypes.c_long(naux), ctypes.c_int(rows), ctypes.c_int(ldx), ctypes.c_longlong(b))
        assert got == old_blk(naux, rows, ldx, b), (naux, rows, ldx, b)

    def pick(*ms):
        return so.PAMD_j2_schedule_pick((ctypes.c_double * 3)(*ms), ctypes.c_int(len(ms)))
    assert pick(108.0, 112.0, 107.5) == 0          # neither challenger wins by 1 %
    assert pick(108.0, 106.0) == 1 and pick(108.0, 106.0, 106.5) == 1
    assert pick(108.0, 112.0, 106.0) == 2 and pick(108.0, 105.0, 103.0) == 2
    assert pick(300.0, 290.0, 289.0) == 1          # fused must beat the CURRENT best by 1 %, not the first candidate


def test_layout_preference_order_is_one_library_function():
    """PAMD_df_layout_pick: packed + full image while 3x fits (2), square rows when only 2x fits (1), packed otherwise (0) - the order
    both df.DF._choose_layout and the C handle's build_rows ask for (each with its own byte counts)."""
    from pyscf_amd import lib
    so = lib.load_library()
    GB = 1 << 30

    def pick(lux, build, after, free, prefer=1):
        return so.PAMD_df_layout_pick(ctypes.c_longlong(lux), ctypes.c_longlong(build), ctypes.c_longlong(after), ctypes.c_longlong(free),
                                      ctypes.c_int(prefer))
    assert pick(234 * GB, 170 * GB, 165 * GB, 287 * GB) == 2            # config 3: all three copies fit
    assert pick(234 * GB, 170 * GB, 165 * GB, 287 * GB, prefer=0) == 1  # ... unless the caller wants ONE copy
    assert pick(391 * GB, 270 * GB, 281 * GB, 287 * GB) == 1            # taxol: 3x does not fit, 2x does
    assert pick(391 * GB, 270 * GB, 290 * GB, 287 * GB) == 0            # ... not with a larger XC reserve afterwards
    assert pick(0, 0, 0, 287 * GB) == 0 and pick(0, 170 * GB, 165 * GB, 287 * GB) == 1
