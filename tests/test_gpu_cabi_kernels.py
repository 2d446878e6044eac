"""Direct C-ABI calls (ctypes + device pointers) of the generic entry points, against numpy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup():
    import torch
    from pyscf_amd import lib
    so = lib.load_library()
    dev = torch.device('cuda', 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return torch, so, dev, st, lib


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize('m,n,k,lower', [(200, 200, 1024, 1), (130, 77, 333, 0), (128, 128, 16, 0), (300, 300, 4096, 1),
                                          (300, 200, 2048, 0), (161, 129, 512, 0), (480, 130, 1024, 0)])   # 160 x 128-tile instance
@pytest.mark.parametrize('glds', [0, 1])
def test_dgemm_tn_split_k_and_lds_dma(m, n, k, lower, glds):
    torch, so, dev, st, lib = _setup()
    rng = np.random.default_rng(m + n + k)
    lda, ldb = (m + 15) // 16 * 16, (n + 15) // 16 * 16
    a = np.zeros((k, lda)); a[:, :m] = rng.standard_normal((k, m))
    b = np.zeros((k, ldb)); b[:, :n] = rng.standard_normal((k, n))
    # slack of 256 doubles after the panels (LDS-DMA kernel reads whole 128-column rows)
    ta = torch.zeros(k * lda + 256, dtype=torch.float64, device=dev); ta[:k * lda] = torch.from_numpy(a.ravel()).to(dev)
    tb = torch.zeros(k * ldb + 256, dtype=torch.float64, device=dev); tb[:k * ldb] = torch.from_numpy(b.ravel()).to(dev)
    if lower:
        tb, ldb, b = ta, lda, a
    nsplit = 3
    part = torch.zeros((nsplit, m, n), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_set_tuning(b'glds', glds))
    flag = lower | 2
    for _ in range(2):            # accumulates
        lib.check(so.PAMD_dgemm_tn(_p(ta), lda, _p(tb), ldb, _p(part), n, m, n, C.c_long(k), flag, nsplit, st))
    lib.check(so.PAMD_set_tuning(b'glds', 1))
    out = torch.empty((m, n), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_reduce_splits(_p(part), nsplit, m, n, _p(out), n, lower, st)) if m == n else None
    want = 2 * a[:, :m].T.dot(b[:, :n])
    if m == n:
        got = out.cpu().numpy()
        if lower:
            assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()      # symmetrised from the lower tiles
        else:
            assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()
    else:
        got = part.sum(0).cpu().numpy()
        assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()


def test_dgemm_nt_pack_unpack_reduce_sym():
    torch, so, dev, st, lib = _setup()
    rng = np.random.default_rng(9)
    m, n, k = 150, 90, 700
    a, b = rng.standard_normal((m, k)), rng.standard_normal((n, k))
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    part = torch.zeros((2, m, n), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_dgemm_nt(_p(ta), C.c_long(k), _p(tb), C.c_long(k), _p(part), n, m, n, C.c_long(k), 2, st))
    assert np.abs(part.sum(0).cpu().numpy() - a.dot(b.T)).max() < 1e-11 * k
    # pack_dm_tril / unpack_tril / reduce_sym
    nao = 37
    dm = rng.standard_normal((2, nao, nao))
    tdm = torch.from_numpy(dm).to(dev)
    npair = nao * (nao + 1) // 2
    tril = torch.zeros((2, npair), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_pack_dm_tril(_p(tdm), 2, nao, _p(tril), st))
    idx = np.tril_indices(nao)
    want = (dm + dm.transpose(0, 2, 1))[:, idx[0], idx[1]]
    want[:, np.arange(nao) * (np.arange(nao) + 1) // 2 + np.arange(nao)] *= .5
    assert np.abs(tril.cpu().numpy() - want).max() < 1e-14
    ld, rows = 48, 40
    full = torch.zeros((2, rows, ld), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_unpack_tril(_p(tril), C.c_long(npair), 2, nao, _p(full), ld, rows, st))
    f = full.cpu().numpy()
    assert np.all(f[:, nao:, :] == 0) and np.all(f[:, :, nao:] == 0)
    assert np.abs(f[:, :nao, :nao] - lib.unpack_tril(want, 1)).max() < 1e-14
    p2 = torch.from_numpy(rng.standard_normal((3, nao, nao))).to(dev)
    out = torch.empty((nao, nao), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_reduce_sym(_p(p2), 3, nao, nao, _p(out), st))
    s = p2.sum(0).cpu().numpy()
    assert np.abs(out.cpu().numpy() - (s + s.T)).max() < 1e-13


def test_error_codes_not_exit():
    """Bad arguments come back as negative codes with a message (the reference's C aborts with exit(1))."""
    torch, so, dev, st, lib = _setup()
    x = torch.zeros(16, dtype=torch.float64, device=dev)
    rc = so.PAMD_df_vj_pass1(_p(x), C.c_long(4), 2, _p(x), 9, _p(x), _p(x), st)
    assert rc < 0 and b'nset' in so.PAMD_last_error()
    rc = so.PAMD_int3c2e_class(0, 1, 0, None, st)
    assert rc < 0
    with pytest.raises(lib.HIPError):
        lib.check(so.PAMD_set_tuning(b'nope', 1))


@pytest.mark.parametrize('m,n,k,nsplit', [(300, 200, 4096, 3), (128, 128, 16 * 2500 * 2, 1), (257, 257, 1024, 4)])
def test_dgemm_tn_masked_and_tile_mask(m, n, k, nsplit):
    """Screened GEMM == dense GEMM with the flagged-off 16 x 128 panel tiles zeroed; the flags come from
    PAMD_tile_mask (16 x 16 tiles), reduced to panels on the host side of the test."""
    torch, so, dev, st, lib = _setup()
    rng = np.random.default_rng(m * 7 + n + k)
    lda, ldb = (m + 15) // 16 * 16, (n + 15) // 16 * 16

    def blocky(rows, ld, cols):
        x = np.zeros((rows, ld))
        x[:, :cols] = rng.standard_normal((rows, cols))
        kill = rng.random((rows // 16, ld // 16)) < 0.6          # 60 % of the 16 x 16 tiles ~ 1e-20
        x *= np.where(np.kron(kill, np.ones((16, 16))) > 0, 1e-20, 1.0)
        return x
    a, b = blocky(k, lda, m), blocky(k, ldb, n)
    ta = torch.zeros(k * lda + 256, dtype=torch.float64, device=dev); ta[:k * lda] = torch.from_numpy(a.ravel()).to(dev)
    tb = torch.zeros(k * ldb + 256, dtype=torch.float64, device=dev); tb[:k * ldb] = torch.from_numpy(b.ravel()).to(dev)
    thr = 1e-15

    def panel_mask(t, ld):
        fl = torch.empty((k // 16, ld // 16), dtype=torch.uint8, device=dev)
        lib.check(so.PAMD_tile_mask(_p(t), C.c_long(ld), C.c_long(k), C.c_double(thr), _p(fl), st))
        f = fl.cpu().numpy()
        ref = np.abs(t[:k * ld].cpu().numpy().reshape(k // 16, 16, ld // 16, 16)).max(axis=(1, 3)) > thr
        assert np.array_equal(f.astype(bool), ref)
        npan = (ld + 127) // 128
        pm = np.zeros((k // 16, npan), np.uint8)
        for j in range(ld // 16):
            pm[:, j // 8] |= f[:, j]
        return pm
    ma, mb = panel_mask(ta, lda), panel_mask(tb, ldb)
    # the kernel indexes masks with ceil(m/128), ceil(n/128) columns
    tm, tn = (m + 127) // 128, (n + 127) // 128
    ma, mb = np.ascontiguousarray(ma[:, :tm]), np.ascontiguousarray(mb[:, :tn])
    tma, tmb = torch.from_numpy(ma).to(dev), torch.from_numpy(mb).to(dev)
    part = torch.zeros((nsplit, m, n), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_dgemm_tn_masked(_p(ta), lda, _p(tb), ldb, _p(part), n, m, n, C.c_long(k), nsplit,
                                      _p(tma), _p(tmb), st))
    got = part.sum(0).cpu().numpy()
    # reference: k-tile contributes to output tile (i, j) only when both panel tiles are flagged
    ref = np.zeros((m, n))
    for i in range(tm):
        for j in range(tn):
            act = (ma[:, i] & mb[:, j]).astype(bool)
            rows = np.repeat(act, 16)
            ref[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128] = \
                a[rows][:, i * 128:min((i + 1) * 128, m)].T @ b[rows][:, j * 128:min((j + 1) * 128, n)]
    assert abs(got - ref).max() < 1e-10
    assert abs(got - a[:, :m].T @ b[:, :n]).max() < 1e-10          # and the skipped work was negligible


@pytest.mark.parametrize('nao,naux,nocc', [(130, 37, 33), (257, 20, 161), (64, 5, 16), (200, 9, 150), (145, 4, 310),
                                           (272, 3, 139), (200, 6, 240), (150, 7, 120), (145, 4, 500), (320, 3, 226),
                                           (145, 5, 200), (200, 4, 216), (272, 3, 360), (130, 2, 176), (145, 3, 100),
                                           (200, 4, 88), (320, 5, 70)])
def test_nr_e2_square_and_fused_rho(nao, naux, nocc):
    """(nocc 240, 120, 500, 226: the 128-orbital instance of the v2 kernel, one to four chunks, with the pair-tail launch;
    r04: nocc 226 / 240 / 200 / 216 / 360 / 176 / 161 end in a last chunk of 7 / 7 / 5 / 6 / 7 / 4 / 4 tiles in the 1 x 4 wave
    arrangement - bitwise the same X as the uniform tiling, tuning key e2wide = 0; nocc 100 / 88 / 70: ONE wide chunk of 7 / 6 / 5.)
    PAMD_unpack_tril -> PAMD_nr_e2_square (LDS-DMA half transform on the unpacked image) against numpy, including the
    first J pass taken from the epilogue: rho_L = sum_{i,p} X[L,i,p] C[p,i] = sum_pq B_L[pq] (C C^T)[pq]."""
    torch, so, dev, st, lib = _setup()
    from pyscf_amd.df import df_jk
    rng = np.random.default_rng(nao + naux)
    npair = nao * (nao + 1) // 2
    tril = rng.standard_normal((naux, npair))
    c = rng.standard_normal((nao, nocc))
    rows = (nao + 15) // 16 * 16
    t_tril = torch.from_numpy(tril).to(dev)
    buf = torch.zeros(naux * rows * rows + 256, dtype=torch.float64, device=dev)
    lib.check(so.PAMD_unpack_tril(_p(t_tril), C.c_long(npair), naux, nao, _p(buf), rows, rows, st))
    sq = buf[:naux * rows * rows].view(naux, rows, rows)
    full = lib.unpack_tril(tril)                                           # (naux, nao, nao) host
    assert np.abs(sq[:, :nao, :nao].cpu().numpy() - full).max() == 0
    orb, nocc_pad, ldo = df_jk.pad_orbitals(c, dev)
    ldx = rows
    x = torch.zeros((naux, nocc_pad, ldx), dtype=torch.float64, device=dev)
    rho = torch.zeros(naux, dtype=torch.float64, device=dev)
    so.PAMD_nr_e2_rho_worksize.restype = C.c_long
    work = torch.zeros(so.PAMD_nr_e2_rho_worksize(naux, ldx, nocc_pad), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_nr_e2_square(_p(sq), C.c_long(rows), rows, naux, nao, _p(orb), ldo, orb.shape[0], nocc_pad,
                                   _p(x), ldx, _p(rho), _p(work), st))
    want = np.einsum('Lpq,qi->Lip', full, c)
    got = x[:, :nocc, :nao].cpu().numpy()
    assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()
    rho_want = np.einsum('Lpq,pq->L', full, c.dot(c.T))
    assert np.abs(rho.cpu().numpy() - rho_want).max() < 1e-10 * np.abs(rho_want).max()
    # without the fused pass (d_rho = NULL) the transform itself is unchanged
    x2 = torch.zeros_like(x)
    lib.check(so.PAMD_nr_e2_square(_p(sq), C.c_long(rows), rows, naux, nao, _p(orb), ldo, orb.shape[0], nocc_pad,
                                   _p(x2), ldx, C.c_void_p(0), C.c_void_p(0), st))
    assert torch.equal(x, x2)
    lib.check(so.PAMD_set_tuning(b'e2wide', 0))
    x3 = torch.zeros_like(x)
    rho3 = torch.zeros(naux, dtype=torch.float64, device=dev)
    try:
        lib.check(so.PAMD_nr_e2_square(_p(sq), C.c_long(rows), rows, naux, nao, _p(orb), ldo, orb.shape[0], nocc_pad,
                                       _p(x3), ldx, _p(rho3), _p(work), st))
    finally:
        lib.check(so.PAMD_set_tuning(b'e2wide', 1))
    assert torch.equal(x, x3)                              # every element: the same k order in either wave arrangement
    assert np.abs(rho3.cpu().numpy() - rho_want).max() < 1e-10 * np.abs(rho_want).max()
    if nocc % 16:
        # r04: nocc rows per aux index in the output (what the K branch uses: the SYRK then contracts naux * nocc rows)
        x4 = torch.full((naux * nocc + 16, ldx), 7.0, dtype=torch.float64, device=dev)
        rho4 = torch.zeros(naux, dtype=torch.float64, device=dev)
        lib.check(so.PAMD_nr_e2_square(_p(sq), C.c_long(rows), rows, naux, nao, _p(orb), ldo, orb.shape[0], nocc,
                                       _p(x4), ldx, _p(rho4), _p(work), st))
        assert torch.equal(x4[:naux * nocc].view(naux, nocc, ldx), x[:, :nocc])
        assert bool((x4[naux * nocc:] == 7.0).all())
        assert np.abs(rho4.cpu().numpy() - rho_want).max() < 1e-10 * np.abs(rho_want).max()
    # the fused pass is deterministic (per-wave partials + fixed-order reduction, no FP atomics): bitwise repeatable,
    # also through the packed-operand kernel
    for fn, args in ((so.PAMD_nr_e2_square, (_p(sq), C.c_long(rows), rows, naux, nao)),
                     (so.PAMD_nr_e2_symm, (_p(t_tril), C.c_long(npair), naux, nao))):
        runs = []
        for _ in range(3):
            r = torch.zeros(naux, dtype=torch.float64, device=dev)
            lib.check(fn(*args, _p(orb), ldo, orb.shape[0], nocc_pad, _p(x2), ldx, _p(r), _p(work), st))
            runs.append(r.clone())
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
        assert np.abs(runs[0].cpu().numpy() - rho_want).max() < 1e-10 * np.abs(rho_want).max()


@pytest.mark.parametrize('nao,naux,nocc', [(200, 9, 150), (145, 4, 310), (272, 3, 139), (100, 3, 150), (1000, 2, 160),
                                           (129, 5, 160), (200, 9, 240), (150, 4, 120), (100, 3, 100), (145, 4, 500),
                                           (320, 3, 226), (145, 5, 200), (200, 4, 216), (272, 3, 360), (130, 2, 176)])
def test_nr_e2_symm_packed_dma_kernel(nao, naux, nocc):
    """PAMD_nr_e2_symm on the shapes that take the all-DMA packed-operand kernel (160- or 128-orbital chunks, whichever pads
    the occupied block less): transposed tiles above
    the diagonal, row tiles below, the crossing tiles visited twice with the keep-masks - against numpy and against the
    register-staged kernel (tuning key pkdma = 0), including the fused first J pass and ragged nao (pad rows / columns)."""
    torch, so, dev, st, lib = _setup()
    from pyscf_amd.df import df_jk
    rng = np.random.default_rng(nao * 7 + nocc)
    npair = nao * (nao + 1) // 2
    tril = rng.standard_normal((naux, npair))
    c = rng.standard_normal((nao, nocc))
    full = lib.unpack_tril(tril)
    want = np.einsum('Lpq,qi->Lip', full, c)
    rho_want = np.einsum('Lpq,pq->L', full, c.dot(c.T))
    t_tril = torch.from_numpy(tril).to(dev)
    orb, nocc_pad, ldo = df_jk.pad_orbitals(c, dev)
    ldx = (nao + 15) // 16 * 16
    so.PAMD_nr_e2_rho_worksize.restype = C.c_long
    work = torch.zeros(max(1, so.PAMD_nr_e2_rho_worksize(naux, ldx, nocc_pad)), dtype=torch.float64, device=dev)
    outs = {}
    for flag in (1, 0):
        lib.check(so.PAMD_set_tuning(b'pkdma', flag))
        x = torch.full((naux, nocc_pad, ldx), 7.0, dtype=torch.float64, device=dev)
        rho = torch.zeros(naux, dtype=torch.float64, device=dev)
        lib.check(so.PAMD_nr_e2_symm(_p(t_tril), C.c_long(npair), naux, nao, _p(orb), ldo, orb.shape[0], nocc_pad, _p(x), ldx,
                                     _p(rho), _p(work), st))
        got = x[:, :nocc, :nao].cpu().numpy()
        assert np.abs(got - want).max() < 1e-11 * np.abs(want).max(), flag
        assert np.abs(rho.cpu().numpy() - rho_want).max() < 1e-10 * np.abs(rho_want).max(), flag
        outs[flag] = got
    lib.check(so.PAMD_set_tuning(b'pkdma', 1))
    assert np.abs(outs[0] - outs[1]).max() < 1e-11 * np.abs(want).max()
    if nocc % 16:
        x4 = torch.full((naux * nocc + 16, ldx), 7.0, dtype=torch.float64, device=dev)     # r04: nocc rows per aux index
        lib.check(so.PAMD_nr_e2_symm(_p(t_tril), C.c_long(npair), naux, nao, _p(orb), ldo, orb.shape[0], nocc, _p(x4), ldx,
                                     C.c_void_p(0), C.c_void_p(0), st))
        got4 = x4[:naux * nocc].view(naux, nocc, ldx)[:, :, :nao].cpu().numpy()
        assert np.abs(got4 - want).max() < 1e-11 * np.abs(want).max()
        assert bool((x4[naux * nocc:] == 7.0).all())
    # r03: the same with the diagonal-block side image (crossing k-tiles read once, no keep-masks)
    so.PAMD_e2_diag_size.restype = C.c_long
    ntile = (ldx + 127) // 128
    assert so.PAMD_e2_diag_size(naux, ldx) == naux * ntile * 128 * 128
    dg = torch.full((naux, ntile, 128, 128), 3.0, dtype=torch.float64, device=dev)
    lib.check(so.PAMD_e2_diag_blocks(_p(t_tril), C.c_long(npair), naux, nao, ldx, _p(dg), st))
    dg_want = np.zeros((naux, ntile * 128, ntile * 128))
    dg_want[:, :nao, :nao] = full
    for t in range(ntile):
        assert np.array_equal(dg[:, t].cpu().numpy(), dg_want[:, t * 128:(t + 1) * 128, t * 128:(t + 1) * 128]), t
    x = torch.full((naux, nocc_pad, ldx), 7.0, dtype=torch.float64, device=dev)
    rho = torch.zeros(naux, dtype=torch.float64, device=dev)
    lib.check(so.PAMD_nr_e2_symm_diag(_p(t_tril), C.c_long(npair), naux, nao, _p(orb), ldo, orb.shape[0], nocc_pad, _p(x), ldx,
                                      _p(rho), _p(work), _p(dg), st))
    got = x[:, :nocc, :nao].cpu().numpy()
    assert np.abs(got - want).max() < 1e-11 * np.abs(want).max()
    assert np.abs(got - outs[1]).max() < 1e-11 * np.abs(want).max()
    assert np.abs(rho.cpu().numpy() - rho_want).max() < 1e-10 * np.abs(rho_want).max()


def test_metric_decompose_cholesky_and_eigen_paths():
    """PAMD_metric_decompose (r04: the ONE factorisation code path of DF.build, Python object and C handle alike;
    pyscf/df/incore.py:150-158, :263-270): M j2c M^T = 1 for the Cholesky form (M lower triangular), the forced
    eigen-decomposition ('ED', df/grad/rhf.py:423-443) and a linearly dependent metric (rank-deficient: fewer rows)."""
    from pyscf_amd import lib
    from pyscf_amd.df import incore
    so = lib.load_library()
    rng = np.random.default_rng(3)
    n = 1100                                                    # above DEVICE_FACTOR_MIN and HOST_EIG_MAX: blocked Cholesky, rocSOLVER syevd
    a = rng.standard_normal((n, n + 50))
    j2c = a.dot(a.T) / n + 0.05 * np.eye(n)
    for force_ed in (0, 1):
        m = np.empty((n, n))
        nrow, tri = C.c_int(), C.c_int()
        lib.check(so.PAMD_metric_decompose(j2c.ctypes.data_as(C.c_void_p), n, C.c_double(1e-7), force_ed, 0, m.ctypes.data_as(C.c_void_p),
                                           C.byref(nrow), C.byref(tri)))
        assert nrow.value == n and tri.value == (0 if force_ed else 1)
        assert np.abs(m.dot(j2c).dot(m.T) - np.eye(n)).max() < 1e-9
        if not force_ed:
            assert np.abs(np.triu(m, 1)).max() == 0.0
            low = np.linalg.cholesky(j2c)
            assert np.abs(m.dot(low) - np.eye(n)).max() < 1e-10
    # the Python layer goes through the same entry point from naux = 1024
    import torch
    mm, tri = incore._decompose_j2c(j2c, 1e-7, 'CD', torch.device('cuda', 0))
    assert tri and np.abs(mm - m0_cd(j2c)).max() < 1e-9
    # linearly dependent: the last 100 functions are copies of the first 100
    j2 = j2c.copy()
    j2[-100:] = j2[:100]
    j2[:, -100:] = j2[:, :100]
    m = np.empty((n, n))
    nrow, tri = C.c_int(), C.c_int()
    lib.check(so.PAMD_metric_decompose(j2.ctypes.data_as(C.c_void_p), n, C.c_double(1e-7), 0, 0, m.ctypes.data_as(C.c_void_p), C.byref(nrow),
                                       C.byref(tri)))
    assert nrow.value == n - 100 and tri.value == 0
    mk = m[:nrow.value]
    assert np.abs(mk.dot(j2).dot(mk.T) - np.eye(nrow.value)).max() < 1e-8


def m0_cd(j2c):
    import scipy.linalg
    low = scipy.linalg.cholesky(j2c, lower=True)
    return scipy.linalg.solve_triangular(low, np.eye(len(low)), lower=True)


def test_balanced_syrk_inside_the_4gib_row_offset_window():
    """the balanced k split makes FULL pieces longer than the uniform ones; the kernels address rows with 32-bit offsets
    inside a 4 GiB buffer window, and a shape like m = 2496 columns x 600 000 rows (uniform piece 200 000 rows = 3.99 GB, balanced
    piece 240 000 rows = 4.8 GB) used to wrap.  The launcher now falls back to uniform pieces there: K = X^T X against torch."""
    torch, so, dev, st, lib = _setup()
    m, k = 2496, 600_000                               # 39 blocks of 64 columns (odd: re-tiled triangle), nsplit 3 by syrk_plan's rule
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.empty((k, m), dtype=torch.float64, device=dev)
    for r0 in range(0, k, 50_000):
        x[r0:r0 + 50_000].normal_(generator=g)
    x *= 1.0 / np.sqrt(k)
    nsplit = 3
    part = torch.zeros((nsplit, m, m), dtype=torch.float64, device=dev)
    lib.check(so.PAMD_dgemm_tn(_p(x), m, _p(x), m, _p(part), m, m, m, C.c_long(k), 1 | 2 | 4 | 8, nsplit, st))
    got = torch.tril(part.sum(dim=0))
    want = torch.zeros((m, m), dtype=torch.float64, device=dev)
    for r0 in range(0, k, 100_000):                    # reference in chunks (a full-size matmul workspace is not needed)
        want += x[r0:r0 + 100_000].T @ x[r0:r0 + 100_000]
    want = torch.tril(want)
    err = float((got - want).abs().max())
    assert err < 1e-11 * float(want.abs().max()), err
