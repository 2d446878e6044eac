"""Cholesky-decomposed 3-centre tensor built on the GPU.

Mirror of ``pyscf/df/incore.py:129-220`` (``cholesky_eri``): j2c = (P|Q) -> Cholesky
(``scipy.linalg.cholesky`` on the host exactly as the reference does at :154, eigen-
decomposition fallback with ``lindep`` :153-158,:263-270) -> per AO-row slab
``cderi[:, slab] = L^-1 (Q|pq)`` (:189-217).  Integrals and the triangular solve run on the
device: ``PAMD_int3c2e_class`` (Rys kernels) and ``PAMD_cderi_solve`` (FP64 MFMA GEMM with the
host-inverted Cholesky factor; rows of this rank's aux shard only).
"""
import ctypes

import numpy as np
import scipy.linalg

from .. import lib as _lib_mod
from ..gto.moleintor import IntEngine, get_engine

LINEAR_DEP_THR = 1e-7   # pyscf/df/incore.py:33


DEVICE_FACTOR_MIN = 1024     # metric size from which the factorisation runs on the GPU


def _decompose_j2c(j2c, lindep, decompose='CD', device=None):
    """-> (M with cderi = M (Q|pq), triangular flag).  M = L^-1 or, when the Cholesky factorisation fails or 'ED' is
    asked for (the decompose_j2c switch of pyscf/df/grad/rhf.py:423-443), (V / sqrt(w))^T over the eigenvalues > lindep.
    From naux = 1024 the Cholesky factor and its triangular inverse are computed on this rank's own GPU (r04: by the library's
    own blocked factorisation, PAMD_metric_decompose - the code PAMD_df_create runs; r02-r03 went through torch's potrf): with one
    process per GPU the eight ranks of a node no longer run eight O(naux^3) LAPACK
    factorisations on the shared host cores at the same time (14 848^3 each at BASELINE config 5), and nothing has to be
    broadcast.  The reference does the same factorisation with scipy on the host (pyscf/df/incore.py:154)."""
    if device is not None and len(j2c) >= DEVICE_FACTOR_MIN:
        # r04: ONE factorisation code path - the library's own blocked Cholesky + block forward substitution on its FP64-MFMA
        # GEMMs (csrc/df_handle.hip: what PAMD_df_create runs), eigen-decomposition inside when a pivot fails or 'ED' is asked for
        import torch
        naux = len(j2c)
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        jh = np.ascontiguousarray(j2c, dtype=np.float64)
        m = np.empty((naux, naux))
        nrow, tri = ctypes.c_int(), ctypes.c_int()
        lib = _lib_mod.load_library()
        _lib_mod.check(lib.PAMD_metric_decompose(jh.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(naux), ctypes.c_double(lindep),
                                                 ctypes.c_int(int(decompose.upper() != 'CD')), ctypes.c_int(idx),
                                                 m.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nrow), ctypes.byref(tri)))
        torch.cuda.set_device(idx)
        return np.ascontiguousarray(m[:nrow.value]), bool(tri.value)
    if decompose.upper() == 'CD':
        try:
            low = scipy.linalg.cholesky(j2c, lower=True)
            linv = scipy.linalg.solve_triangular(low, np.eye(len(low)), lower=True, check_finite=False)
            return linv, True
        except scipy.linalg.LinAlgError:
            pass
    w, v = scipy.linalg.eigh(j2c)
    mask = w > lindep
    v = v[:, mask] / np.sqrt(w[mask])
    return v.T, False


def cholesky_eri_gpu(mol, auxmol, device, l0=None, l1=None, lindep=LINEAR_DEP_THR,
                     slab_bytes=24 << 30, engine=None, return_engine=False, omega=0.0, decompose_j2c='CD', layout='packed'):
    """Rows [l0, l1) of cderi as a torch CUDA tensor: (nL, nao_pair) packed like the reference's `_cderi`
    (pyscf/df/df.py:59-72), or - layout='square', r06 - (nL, rows, rows) with both triangles of every B_L (rows =
    round_up(nao, 16), pads zero): every column slab is solved into a work buffer and scattered into the square rows
    (PAMD_unpack_tril_slab), the packed tensor never exists."""
    import torch
    lib = _lib_mod.load_library()
    eng = engine or get_engine(mol, auxmol, device, omega)
    naux = eng.aux.nao
    nao = eng.ao.nao
    npair = nao * (nao + 1) // 2
    j2c = eng.int2c2e().cpu().numpy()
    j2c = (j2c + j2c.T) * .5
    M, tri = _decompose_j2c(j2c, lindep, decompose_j2c, device)
    nrow_total = M.shape[0]
    if l0 is None:
        l0, l1 = 0, nrow_total
    l1 = min(l1, nrow_total)
    nL = max(l1 - l0, 0)
    # M^T columns of this shard: [naux][nL]
    lda = max((nL + 15) // 16 * 16, 16)
    mt = np.zeros((naux, lda))
    mt[:, :nL] = M[l0:l1].T
    mt_dev = torch.from_numpy(mt).to(device)
    if torch.device(device).type == 'cuda':
        free = (torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
        rows_sq = (nao + 15) // 16 * 16
        need = (nL * npair * 8 if layout != 'square' else nL * rows_sq * rows_sq * 8 + min(slab_bytes, npair * nL * 8)) + \
            min(slab_bytes, npair * naux * 8)
        if need > free:
            raise MemoryError('DF tensor shard of %d x %d doubles (%.1f GB + %.1f GB of slab workspace) does not fit the %.1f GB '
                              'free on %s: shard the auxiliary index over more ranks (one process per GPU), the out-of-core '
                              'contraction of pyscf/df/outcore.py is not restated'
                              % (nL, npair, nL * npair * 8e-9, min(slab_bytes, npair * naux * 8) * 1e-9, free * 1e-9, device))
    square = layout == 'square'
    if square:
        rows_sq = (nao + 15) // 16 * 16
        from .df import DF as _DF
        cderi = _DF.alloc_square(nL, rows_sq, device)          # zeroed, padded aux-row stride (DF.SQ_STRIDE_PAD)
    else:
        cderi = torch.empty((nL, npair), dtype=torch.float64, device=device)
    # AO row-shell slabs bounded by slab_bytes of T = [rows][naux]
    nsh = eng.ao.n
    max_rows = max(int(slab_bytes // (naux * 8)), 1)
    slabs = []
    sh0 = 0
    while sh0 < nsh:
        sh1 = sh0 + 1
        while sh1 < nsh:
            r0, r1 = eng.slab_rows(sh0, sh1 + 1)
            if r1 - r0 > max_rows:
                break
            sh1 += 1
        slabs.append((sh0, sh1))
        sh0 = sh1
    bufrows = max(eng.slab_rows(a, b)[1] - eng.slab_rows(a, b)[0] for a, b in slabs)
    T = torch.empty((bufrows, naux), dtype=torch.float64, device=device)
    slab_out = torch.empty((max(nL, 1) * bufrows,), dtype=torch.float64, device=device) if square else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for sh0, sh1 in slabs:
        r0, r1 = eng.slab_rows(sh0, sh1)
        Tv = eng.int3c2e_slab(sh0, sh1, out=T)
        if square:
            # the slab's packed columns [r0, r1) of every aux row, ld = r1 - r0, then both triangles of the square rows
            ncol = r1 - r0
            p0 = int(eng.ao.ao0[sh0])
            p1 = int(eng.ao.ao0[sh1]) if sh1 < eng.ao.n else nao
            _lib_mod.check(lib.PAMD_cderi_solve(
                ctypes.c_void_p(mt_dev.data_ptr()), ctypes.c_int(lda),
                ctypes.c_void_p(Tv.data_ptr()), ctypes.c_long(naux),
                ctypes.c_void_p(slab_out.data_ptr()), ctypes.c_long(ncol),
                ctypes.c_int(nL), ctypes.c_long(ncol), ctypes.c_int(naux), ctypes.c_int(l0),
                ctypes.c_int(1 if tri else 0), st))
            _lib_mod.check(lib.PAMD_unpack_tril_slab(
                ctypes.c_void_p(slab_out.data_ptr()), ctypes.c_long(ncol), ctypes.c_int(nL), ctypes.c_int(p0), ctypes.c_int(p1),
                ctypes.c_void_p(cderi.data_ptr()), ctypes.c_int(rows_sq), ctypes.c_long(cderi.stride(0)), st))
            continue
        _lib_mod.check(lib.PAMD_cderi_solve(
            ctypes.c_void_p(mt_dev.data_ptr()), ctypes.c_int(lda),
            ctypes.c_void_p(Tv.data_ptr()), ctypes.c_long(naux),
            ctypes.c_void_p(cderi.data_ptr() + 8 * r0), ctypes.c_long(npair),
            ctypes.c_int(nL), ctypes.c_long(r1 - r0), ctypes.c_int(naux), ctypes.c_int(l0),
            ctypes.c_int(1 if tri else 0), st))
    torch.cuda.synchronize()
    if return_engine:
        return cderi, eng
    return cderi


def aux_e2_gpu(mol, auxmol, device, omega=0.0):
    """(naux, nao_pair) s2ij tensor of raw 3-centre integrals (df.incore.aux_e2 analogue,
    pyscf/df/incore.py:40-70), transposed to the cderi layout; for tests."""
    eng = IntEngine(mol, auxmol, device, omega)
    T = eng.int3c2e_slab(0, eng.ao.n)
    return T.T.contiguous()
