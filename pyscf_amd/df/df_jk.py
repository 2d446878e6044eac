"""Density-fitted J/K build on MI355X.

Host-side mirror of ``pyscf/df/df_jk.py:280-413`` (``get_jk``): same arguments, same
branches (J packed-tril two-pass product :329-337,:367; K MO branch :339-381; K general-DM
branch :382-408; complex DMs handled by real/imag split as ``_DFHF.get_jk`` does at
:160-171).  The numerical work is done by hand-written gfx950 kernels reached through the
C ABI of ``libpyscf_amd.so`` (include/pyscf_amd.h); there is no CPU fallback.

Device data: ``dfobj._cderi_dev`` = the rank-local row shard ``cderi[l0:l1, :nao_pair]``
(torch CUDA tensor, f64), laid out exactly like the reference's ``_cderi``.
With ``torch.distributed`` initialised and world_size > 1 the partial J/K of the aux-index
shards are summed by an RCCL all-reduce.
"""
import ctypes

import numpy as np

from .. import lib as _lib_mod
from ..lib import comm as _comm

_c = ctypes


def _torch():
    import torch
    return torch


def _ptr(t):
    return _c.c_void_p(t.data_ptr())


def _stream():
    return _c.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _round_up(x, m):
    return (x + m - 1) // m * m


class KernelTimer:
    """Optional per-kernel timing with HIP events recorded on the launch stream (torch's current
    stream is the stream every PAMD_* launch uses).  bench.py reads `.summary()`."""

    def __init__(self):
        self.records = []

    def call(self, name, fn, *args):
        torch = _torch()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.records.append((name, e0, e1))
        return rc

    def summary(self):
        _torch().cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            tot, cnt = out.get(name, (0.0, 0))
            out[name] = (tot + ms, cnt + 1)
        return out

    def reset(self):
        self.records = []


def _call(dfobj, name, fn, *args):
    timer = getattr(dfobj, 'kernel_timer', None)
    rc = timer.call(name, fn, *args) if timer is not None else fn(*args)
    _lib_mod.check(rc)


def _allreduce(dfobj, tensors):
    """Sum the rank-partial results over the aux-index shards (RCCL over xGMI)."""
    if getattr(dfobj, '_shard_override', None) is None:
        _comm.all_reduce(tensors, dfobj.group, dfobj.world_size)


def _allreduce_jk_packed(dfobj, lib, vjtril, vk):
    """One collective for [J~ (packed) || K (packed)] when K is symmetric (MO branch): 2 nao_pair doubles per density instead
    of nao_pair + nao^2 (27.6 MB instead of 41.3 MB at config 3).  K is packed as (K + K^T)(1 - delta_pq / 2) by the kernel
    that packs density matrices, summed over the ranks, unpacked by PAMD_unpack_tril and halved off the diagonal."""
    torch = _torch()
    st = _stream()
    nset, nao = vk.shape[0], vk.shape[-1]
    npair = nao * (nao + 1) // 2
    nj = vjtril.shape[0]
    buf = torch.empty((nj + nset, npair), dtype=torch.float64, device=vk.device)
    buf[:nj].copy_(vjtril)
    _call(dfobj, 'pack_dm_tril', lib.PAMD_pack_dm_tril, _ptr(vk), _c.c_int(nset), _c.c_int(nao), _ptr(buf[nj:]), st)
    _comm.all_reduce([buf], dfobj.group, dfobj.world_size)
    vjtril.copy_(buf[:nj])
    _call(dfobj, 'unpack_tril', lib.PAMD_unpack_tril, _ptr(buf[nj:]), _c.c_long(npair), _c.c_int(nset), _c.c_int(nao), _ptr(vk),
          _c.c_int(nao), _c.c_int(nao), st)
    vk.mul_(0.5)
    vk.diagonal(dim1=-2, dim2=-1).mul_(2.0)


def _square_rows(dfobj):
    """The square rows sq[L][rows][ld] when they ARE the tensor (DF.layout 'square', r06), else None."""
    return dfobj._cderi_sq if getattr(dfobj, '_layout', None) == 'square' else None


def _sq_args(sq, b0=0, nb=None):
    """(pointer, lstride, ld) of the square rows [b0, b0 + nb) for the PAMD_*_sq entry points."""
    return _ptr(sq[b0:] if nb is None else sq[b0:b0 + nb]), _c.c_long(sq.stride(0)), _c.c_int(sq.shape[2])


def _vj_pass2_rows(dfobj, lib, b0, nb, nao, rho_ptr, ns, vj_ptr, st):
    """vjtril[s][pq] += sum_{L in [b0, b0 + nb)} rho[s][L] B[L, pq] from whichever layout holds the rows."""
    sq = _square_rows(dfobj)
    if sq is not None:
        p, lstride, ld = _sq_args(sq, b0, nb)
        _call(dfobj, 'vj_pass2', lib.PAMD_df_vj_pass2_sq, p, lstride, ld, _c.c_int(nao), _c.c_int(nb), rho_ptr, _c.c_int(ns), vj_ptr, st)
    else:
        cderi = dfobj._packed
        _call(dfobj, 'vj_pass2', lib.PAMD_df_vj_pass2, _ptr(cderi[b0:b0 + nb]), _c.c_long(cderi.shape[1]), _c.c_int(nb), rho_ptr,
              _c.c_int(ns), vj_ptr, st)


def _vj_pass1(dfobj, lib, dms_dev, nset, nao):
    """rho[s][L] = sum_pq B[L,pq] dtril[s][pq] on the current stream; returns the state _vj_pass2 needs."""
    torch = _torch()
    naux, npair = dfobj.tensor_shape()
    st = _stream()
    dev = dfobj.tensor_device()
    sq = _square_rows(dfobj)
    rhos = []
    for s0 in range(0, nset, 4):
        ns = min(4, nset - s0)
        dmtril = torch.empty((ns, npair), dtype=torch.float64, device=dev)
        _call(dfobj, 'pack_dm_tril', lib.PAMD_pack_dm_tril, _ptr(dms_dev[s0:s0 + ns]), _c.c_int(ns), _c.c_int(nao),
                                             _ptr(dmtril), st)
        rho = torch.empty((ns, naux), dtype=torch.float64, device=dev)
        wlen = lib.PAMD_df_vj_pass1_worksize(_c.c_long(npair), _c.c_int(naux), _c.c_int(ns))
        work = torch.empty((max(wlen, 1),), dtype=torch.float64, device=dev)
        if sq is not None:
            p, lstride, ld = _sq_args(sq)
            _call(dfobj, 'vj_pass1', lib.PAMD_df_vj_pass1_sq, p, lstride, ld, _c.c_int(nao), _c.c_int(naux), _ptr(dmtril), _c.c_int(ns),
                  _ptr(rho), _ptr(work), st)
        else:
            _call(dfobj, 'vj_pass1', lib.PAMD_df_vj_pass1, _ptr(dfobj._packed), _c.c_long(npair), _c.c_int(naux),
                                                _ptr(dmtril), _c.c_int(ns), _ptr(rho), _ptr(work), st)
        rhos.append((s0, ns, rho, dmtril, work))
    return rhos


def _vj_pass2(dfobj, lib, rhos, nset):
    """vjtril[s][pq] = sum_L rho[s][L] B[L,pq] on the current stream."""
    torch = _torch()
    naux, npair = dfobj.tensor_shape()
    nao = int((np.sqrt(8.0 * npair + 1) - 1) / 2 + .5)
    st = _stream()
    vjtril = torch.zeros((nset, npair), dtype=torch.float64, device=dfobj.tensor_device())
    for s0, ns, rho, _dmtril, _work in rhos:
        _vj_pass2_rows(dfobj, lib, 0, naux, nao, _ptr(rho), ns, _ptr(vjtril[s0:s0 + ns]), st)
    return vjtril


def _vj(dfobj, lib, dms_dev, nset, nao):
    return _vj_pass2(dfobj, lib, _vj_pass1(dfobj, lib, dms_dev, nset, nao), nset)


def _k_blocksize(dfobj, naux, rows, ldx):
    """aux rows per half-transform block: X block (blk*rows*ldx*8 B) sized from max_memory
    like the reference's `blksize` (df_jk.py:359-360), but against HBM (default 16 GB)."""
    so = _lib_mod.load_library()             # r06: the library's rule (PAMD_k_block_rows; the C handle asks the same function)
    so.PAMD_k_block_rows.restype = _c.c_long
    return int(so.PAMD_k_block_rows(_c.c_long(int(naux)), _c.c_int(int(rows)), _c.c_int(int(ldx)), _c.c_longlong(int(dfobj.k_block_bytes))))


def _rho_work(dfobj, lib, nb, ldx, nocc_pad):
    """Per-wave partials of the fused first J pass (summed in a fixed order: J is bitwise reproducible)."""
    n = lib.PAMD_nr_e2_rho_worksize(_c.c_int(nb), _c.c_int(ldx), _c.c_int(nocc_pad))
    return _ptr(dfobj._workspace('rho_work', (max(int(n), 1),)))


def syrk_items(nao):
    """Work items of the re-tiled SYRK triangle (PAMD_syrk_item_count): 0 when the 2 x 2 tiling is kept (even number of 64-column
    blocks, small matrices)."""
    return int(_lib_mod.load_library().PAMD_syrk_item_count(_c.c_int(int(nao))))


def syrk_plan(nao, nsplit=None, flags=None, reserve=0):
    """(flags, nsplit) of the K = X^T X product (flag 1: lower triangle, flag 2: LDS-DMA operands) - r06: from the library's
    PAMD_syrk_plan, the ONE copy of the rule (the C handle calls the same function; the Python transcription is gone).

    Default when the matrix has an odd number of 64-column blocks (nao = 1856: 29): the RE-TILED triangle (flag 8,
    syrk_slots_kernel: work items of four live 64 x 64 wave blocks, 110 items instead of 120 tiles with 45 dead wave blocks)
    with the BALANCED k split (flag 4): n full pieces of K m / (n m + 1) rows per item plus one 1/m-length remainder piece that
    runs in the workgroup slots the full pieces leave free - n = 4, m = 2 at nao = 1856: 440 + 55 of the 512 slots, 4.5 effective
    splits instead of 4.  Measured (profiles/r03/kbench_syrk_variants.log, K only): uniform 39.7 ms, balanced alone 41.2, re-tiled
    alone 39.6, both 35.4.  Otherwise 128 x 128 tiles and 4 uniform splits (120 x 4 = 480 slots, one round).  Earlier variants
    that lost (profiles/r02): 17 uniform splits (46.4 vs 44.0 ms), 160 x 128 tiles x 5 splits (42.0 vs 41.6), stream-K (41.5 vs 39.6)."""
    f, n = _c.c_int(), _c.c_int()
    _lib_mod.check(_lib_mod.load_library().PAMD_syrk_plan(_c.c_int(int(nao)), _c.c_int(int(reserve)), _c.c_int(-1 if flags is None else int(flags)),
                                                          _c.c_int(int(nsplit or 0)), _c.byref(f), _c.byref(n)))
    return f.value, n.value


def orbital_ld(nocc_pad):
    """Leading dimension of a zero-padded orbital operand with nocc_pad columns: whole chunks of every half-transform kernel
    (the library's rule, PAMD_e2_orb_ld in df_jk.hip)."""
    return int(_lib_mod.load_library().PAMD_e2_orb_ld(_c.c_int(int(nocc_pad))))


def pad_orbitals(orbo, device):
    """Host (nao, nocc) occupied-orbital block C_occ*sqrt(occ) -> zero-padded device operand
    (orb[nao][ldo], nocc_pad) in the layout PAMD_nr_e2_symm expects."""
    torch = _torch()
    nao, nocc = orbo.shape
    nocc_pad = _round_up(max(nocc, 1), 16)
    # whole chunks of the exact-tile kernels (32 wa columns), of the v2 DMA kernels (160 or 128 columns, df_jk.hip::v2_tile) and of
    # the 128-column chunks with a narrower last one (v2_wide)
    ldo = orbital_ld(nocc_pad)
    orb_h = np.zeros((_round_up(nao, 16), ldo))          # zero rows up to a multiple of the k-tile
    orb_h[:nao, :nocc] = orbo
    orb = torch.from_numpy(orb_h).to(device)
    orb.norb = nocc                                      # the K branch keeps `norb` rows per aux index in X (r04), not nocc_pad
    return orb, (nocc_pad if nocc else 0), ldo


def _vk_mo(dfobj, lib, orb_list, nao, after_e2=None, fuse_j=None, j_corun=True, j_fused=None):
    """K_pq = sum_{L,i} X[L,i,p] X[L,i,q],  X[L,i,p] = sum_q B_L[p,q] orbo[q,i]
    (df_jk.py:353-380; nr_ao2mo.c:399-419,1240-1266).  orb_list: [(orb_dev, nocc_pad, ldo)]."""
    torch = _torch()
    naux, npair = dfobj.tensor_shape()
    dev = dfobj.tensor_device()
    st = _stream()
    ldx = _round_up(nao, 16)
    if j_fused is not None and _square_rows(dfobj) is not None:
        raise RuntimeError('the second J pass inside the SYRK kernel (PAMD_syrk_jfused) reads packed rows: not a schedule of the square layout')
    cderi = dfobj._packed                       # None in the square layout (every row then has its square form)
    kflags = getattr(dfobj, 'k_syrk_flags', None)
    reserve = 0
    if j_fused is not None:
        j_corun = False                      # r05: the second J pass rides INSIDE the SYRK kernel (PAMD_syrk_jfused): full balanced grid
    if kflags is None and after_e2 is not None and j_corun:
        # the second J pass runs beside this SYRK on the side stream: it hides in the 32 workgroup slots the plain 120 x 4 grid
        # leaves idle (J/K 108.8 ms) but not beside the balanced schedule that fills them (110.0 ms; K alone: 35.6 vs 39.7 ms): the
        # SYRK's own L2 -> LDS panel traffic and the J stream share one path - plain grid when J co-runs,
        # or (DF.k_syrk_reserve > 0, late r04) the balanced re-tiled schedule sized to leave that many slots to the pass
        reserve = int(getattr(dfobj, 'k_syrk_reserve', 0) or 0)
        if not reserve:
            kflags = 0
    syrk_flags, nsplit = syrk_plan(nao, dfobj.k_nsplit, kflags, reserve)
    if reserve and (syrk_flags & 4):
        syrk_flags |= (reserve // 4) << 8          # PAMD_dgemm_tn: bits 8-15 of the flags = slots the balanced split leaves free, / 4
    vks = []
    for iset, (orb, nocc_pad, ldo) in enumerate(orb_list):
        vk = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
        if nocc_pad == 0 or naux == 0:
            vks.append(vk)
            continue
        blk = _k_blocksize(dfobj, naux, nocc_pad, ldx)
        if after_e2 is not None and fuse_j is None and naux >= 2:
            blk = min(blk, -(-naux // 2))      # two-pass J beside K: pass 1 behind the first block's SYRK, pass 2 behind the second's
        # rows per aux index in X: the orbitals themselves where the operand says how many there are (pad_orbitals), so that the
        # SYRK contracts nb * norb rows - zero rows pad the END of a block to a whole k-tile, not every aux index (nocc = 226: 6 %)
        xr = int(getattr(orb, 'norb', 0)) or nocc_pad
        X = dfobj._workspace('X', (blk * nocc_pad + 16, ldx))
        part = dfobj._workspace('kpart', (nsplit, nao, nao))
        part.zero_()
        sq = dfobj.square_image() if hasattr(dfobj, 'square_image') else None
        nsq = sq.shape[0] if sq is not None else 0
        bounds = list(range(0, naux, blk)) + [naux]
        if 0 < nsq < naux:
            # partial image (rows [0, nsq)): cut the K blocks at its end
            bounds = sorted(set(list(range(0, nsq, blk)) + [nsq] + list(range(nsq, naux, blk)) + [naux]))
        # e2-first order inside a block (r03): the half transforms of its `nsub` sub-blocks are queued back to back and the second
        # J pass of sub-block s (side stream) runs beside the half transform of sub-block s + 1 - a pure HBM stream hides ~80 %
        # behind that kernel (its operands arrive by DMA one tile ahead) but only ~25 % behind the SYRK, whose L2 -> LDS panel
        # traffic shares the path with it; only the last sub-block's pass is left for the SYRK.  One SYRK per block as before.
        nsub = max(1, int(getattr(dfobj, 'k_e2_pipeline', 1) or 1)) if after_e2 is not None else 1
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            nb = b1 - b0
            cuts = sorted(set(b0 + (nb * s_) // nsub for s_ in range(nsub + 1)))
            for s0, s1 in zip(cuts[:-1], cuts[1:]):
                ns, xs = s1 - s0, X[(s0 - b0) * xr:]
                if b1 <= nsq:
                    # fuse_j[set] = rho (naux) zeroed: the first J pass of the density orb orb^T comes out of the epilogue
                    rho_j = fuse_j[iset] if fuse_j is not None else None
                    _call(dfobj, 'e2_symm', lib.PAMD_nr_e2_square_ls, _ptr(sq[s0:s0 + ns]), _c.c_long(sq.shape[2]),
                          _c.c_int(sq.shape[1]), _c.c_long(sq.stride(0)), _c.c_int(ns), _c.c_int(nao), _ptr(orb), _c.c_int(ldo),
                          _c.c_int(orb.shape[0]), _c.c_int(xr), _ptr(xs), _c.c_int(ldx),
                          _ptr(rho_j[s0:]) if rho_j is not None else _c.c_void_p(0),
                          _rho_work(dfobj, lib, ns, ldx, nocc_pad) if rho_j is not None else _c.c_void_p(0), st)
                else:
                    _e2_packed(dfobj, lib, s0, ns, nao, orb, ldo, xr, xs, ldx,
                               _ptr(fuse_j[iset][s0:]) if fuse_j is not None else None,
                               _rho_work(dfobj, lib, ns, ldx, nocc_pad) if fuse_j is not None else None, st)
                if after_e2 is not None and j_fused is None:
                    after_e2(s0, ns, iset)
            kx = nb * xr
            kx16 = _round_up(kx, 16)
            if kx16 > kx:
                X[kx:kx16].zero_()
            if j_fused is not None:
                # vj[pq] += sum_L rho[L] B[L][pq] for the rows [b0, b1) inside the SYRK kernel of the same rows; 1 = no fused form for
                # this shape (nothing launched): the pass in line, then the plain call
                vj_f = j_fused
                box = {}

                def fused_call(*a):
                    box['rc'] = lib.PAMD_syrk_jfused(*a)
                    return 0 if box['rc'] == 1 else box['rc']
                _call(dfobj, 'dgemm_tn', fused_call, _ptr(X), _c.c_int(ldx), _ptr(part), _c.c_int(nao), _c.c_int(nao),
                      _c.c_long(kx16), _c.c_int(syrk_flags), _c.c_int(nsplit), _ptr(cderi[b0:b1]), _c.c_long(npair),
                      _c.c_int(nb), _ptr(fuse_j[iset][b0:]), _ptr(vj_f[iset]), st)
                rc = box['rc']
                if rc == 0:
                    continue
                _vj_pass2_rows(dfobj, lib, b0, nb, nao, _ptr(fuse_j[iset][b0:]), 1, _ptr(vj_f[iset]), st)
            _call(dfobj, 'dgemm_tn', lib.PAMD_dgemm_tn, _ptr(X), _c.c_int(ldx), _ptr(X), _c.c_int(ldx), _ptr(part),
                  _c.c_int(nao), _c.c_int(nao), _c.c_int(nao), _c.c_long(kx16), _c.c_int(syrk_flags),
                  _c.c_int(nsplit), st)
        _call(dfobj, 'reduce_splits', lib.PAMD_reduce_splits, _ptr(part), _c.c_int(nsplit), _c.c_int(nao),
              _c.c_int(nao), _ptr(vk), _c.c_int(nao), _c.c_int(1), st)
        vks.append(vk)
    return torch.stack(vks)


def _e2_packed(dfobj, lib, b0, nb, nao, orb, ldo, nocc_pad, out, ldx, rho, rho_work, st):
    """Half transform of the packed rows [b0, b0 + nb) (PAMD_nr_e2_symm), with the diagonal-block side image when the tensor
    object keeps one for these rows (DF.diag_image, ldx = round_up(nao, 16))."""
    cderi = dfobj._packed
    dg, row0 = dfobj.diag_image() if hasattr(dfobj, 'diag_image') and ldx == _round_up(nao, 16) else (None, 0)
    null = _c.c_void_p(0)
    if dg is not None and b0 >= row0:
        _call(dfobj, 'e2_symm', lib.PAMD_nr_e2_symm_diag, _ptr(cderi[b0:b0 + nb]), _c.c_long(cderi.shape[1]), _c.c_int(nb),
              _c.c_int(nao), _ptr(orb), _c.c_int(ldo), _c.c_int(orb.shape[0]), _c.c_int(nocc_pad), _ptr(out), _c.c_int(ldx),
              rho if rho is not None else null, rho_work if rho is not None else null, _ptr(dg[b0 - row0:]), st)
    else:
        _call(dfobj, 'e2_symm', lib.PAMD_nr_e2_symm, _ptr(cderi[b0:b0 + nb]), _c.c_long(cderi.shape[1]), _c.c_int(nb),
              _c.c_int(nao), _ptr(orb), _c.c_int(ldo), _c.c_int(orb.shape[0]), _c.c_int(nocc_pad), _ptr(out), _c.c_int(ldx),
              rho if rho is not None else null, rho_work if rho is not None else null, st)


def _vk_general(dfobj, lib, dms_dev, nset, nao):
    """vk = einsum('pki,pkj->ij', einsum('pij,jk->pki', B, D), B)   (df_jk.py:382-407)."""
    torch = _torch()
    cderi = dfobj._packed
    naux, npair = dfobj.tensor_shape()
    dev = dfobj.tensor_device()
    st = _stream()
    ldx = _round_up(nao, 16)
    rows = _round_up(nao, 16)
    ldo = _round_up(rows, 160) if rows > 160 else _round_up(rows, 32)     # whole 32-column wave tiles (square-image kernel)
    ldo = max(ldo, orbital_ld(rows))
    nsplit = dfobj.k_nsplit or 4
    blk = max(1, _k_blocksize(dfobj, naux, rows, ldx) // 2)
    vk = torch.zeros((nset, nao, nao), dtype=torch.float64, device=dev)
    if naux == 0:
        return vk
    X = dfobj._workspace('X', (blk, rows, ldx))
    # rows with an unpacked image: it IS the second operand (no per-block unpack) and feeds the square-image half transform
    sq = dfobj.square_image() if hasattr(dfobj, 'square_image') else None
    nsq = sq.shape[0] if sq is not None and sq.shape[1] == rows and sq.shape[2] == ldx else 0
    bounds = sorted(set(list(range(0, nsq, blk)) + [nsq] + list(range(nsq, naux, blk)) + [naux]))
    full = None
    padded = sq is not None and nsq and sq.stride(0) != rows * ldx       # square LAYOUT: padded aux-row stride (DF.SQ_STRIDE_PAD)
    if nsq < naux or padded:
        full = dfobj._workspace('full', (blk, rows, ldx))
        full.zero_()
    for k in range(nset):
        orb = torch.zeros((rows, ldo), dtype=torch.float64, device=dev)
        orb[:nao, :nao] = dms_dev[k]
        part = torch.zeros((nsplit, nao, nao), dtype=torch.float64, device=dev)
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            nb = b1 - b0
            if nb == 0:
                continue
            if b1 <= nsq:
                _half_transform(dfobj, lib, b0, nb, orb, rows, ldo, nao, X, ldx, st)
                second = sq[b0:b1]
                if padded:
                    # the product below reads the rows of consecutive aux indices as ONE tall matrix [nb rows][ldx]: a contiguous copy
                    # of the block (a device copy against 4 nb nao^3 flops; the general-DM branch is the rare one, df_jk.py:382-407)
                    full[:nb].copy_(second)
                    second = full
            else:
                _e2_packed(dfobj, lib, b0, nb, nao, orb, ldo, rows, X, ldx, None, None, st)
                _call(dfobj, 'unpack_tril', lib.PAMD_unpack_tril, _ptr(cderi[b0:b1]), _c.c_long(npair), _c.c_int(nb), _c.c_int(nao),
                      _ptr(full), _c.c_int(ldx), _c.c_int(rows), st)
                second = full
            _call(dfobj, 'dgemm_tn', lib.PAMD_dgemm_tn, _ptr(X), _c.c_int(ldx), _ptr(second), _c.c_int(ldx), _ptr(part),
                  _c.c_int(nao), _c.c_int(nao), _c.c_int(nao), _c.c_long(nb * rows), _c.c_int(0 | 2), _c.c_int(nsplit), st)
        _call(dfobj, 'reduce_splits', lib.PAMD_reduce_splits, _ptr(part), _c.c_int(nsplit), _c.c_int(nao), _c.c_int(nao),
              _ptr(vk[k]), _c.c_int(nao), _c.c_int(0), st)
    return vk


def _half_transform(dfobj, lib, b0, nb, orb, nocc_pad, ldo, nao, out, ldx, st):
    """out[L][i][p] = sum_q B_L[p,q] orb[q,i] for aux rows [b0, b0 + nb): square-image kernel when the image exists."""
    sq = dfobj.square_image() if hasattr(dfobj, 'square_image') else None
    if sq is not None and b0 + nb <= sq.shape[0]:
        _call(dfobj, 'e2_symm', lib.PAMD_nr_e2_square_ls, _ptr(sq[b0:b0 + nb]), _c.c_long(sq.shape[2]), _c.c_int(sq.shape[1]),
              _c.c_long(sq.stride(0)), _c.c_int(nb), _c.c_int(nao), _ptr(orb), _c.c_int(ldo), _c.c_int(orb.shape[0]), _c.c_int(nocc_pad), _ptr(out),
              _c.c_int(ldx), _c.c_void_p(0), _c.c_void_p(0), st)
    else:
        _e2_packed(dfobj, lib, b0, nb, nao, orb, ldo, nocc_pad, out, ldx, None, None, st)


def _vk_lowrank(dfobj, lib, lefts, rights, sym, nao):
    """Exchange of factorised densities D_k = L_k R_k^T (+ R_k L_k^T when sym): K(D_k) = sum_L (B_L L_k)(B_L R_k)^T, two
    MO-branch half transforms and one full X^T Y product per density - 6 naux nao^2 r flops instead of the 4 naux nao^3 of
    the general branch (pyscf/df/df_jk.py:382-407).  This is what a TDA / TDDFT / CPHF / Newton trial density
    C_occ x C_vir^T (rank nocc) needs (pyscf/scf/_response_functions.py:29-247 feeds get_jk with such matrices); densities
    that share their left factor (the occupied orbitals of one reference state) share its half transform."""
    torch = _torch()
    naux = dfobj.tensor_shape()[0]
    dev = dfobj.tensor_device()
    st = _stream()
    nset = len(lefts)
    ldx = _round_up(nao, 16)
    nsplit = dfobj.k_nsplit or 4
    vk = torch.zeros((nset, nao, nao), dtype=torch.float64, device=dev)
    if naux == 0:
        return vk
    # group the densities by the identity of their left factor
    groups = []
    for k in range(nset):
        if groups and groups[-1][0] is lefts[k]:
            groups[-1][1].append(k)
        else:
            groups.append((lefts[k], [k]))
    rpad_max = max(_round_up(max(l.shape[1], 1), 16) for l in lefts)
    blk = max(1, _k_blocksize(dfobj, naux, rpad_max, ldx) // 2)
    X = dfobj._workspace('X', (blk, rpad_max, ldx))
    Y = dfobj._workspace('Y', (blk, rpad_max, ldx))
    for lf, members in groups:
        r = lf.shape[1]
        if r == 0:
            continue
        orb_l, rpad, ldo = pad_orbitals(np.ascontiguousarray(lf, dtype=np.float64), dev)
        orbs_r = [pad_orbitals(np.ascontiguousarray(rights[k], dtype=np.float64), dev) for k in members]
        assert all(o[1] == rpad for o in orbs_r), 'left and right factors need the same number of columns'
        Xv = X.view(-1)[:blk * rpad * ldx].view(blk, rpad, ldx)
        Yv = Y.view(-1)[:blk * rpad * ldx].view(blk, rpad, ldx)
        parts = dfobj._workspace('kpart_lr', (len(members), nsplit, nao, nao))
        parts.zero_()
        for b0 in range(0, naux, blk):
            nb = min(blk, naux - b0)
            _half_transform(dfobj, lib, b0, nb, orb_l, rpad, ldo, nao, Xv, ldx, st)      # shared by the group
            for ik in range(len(members)):
                orb_r, _, ldo_r = orbs_r[ik]
                _half_transform(dfobj, lib, b0, nb, orb_r, rpad, ldo_r, nao, Yv, ldx, st)
                _call(dfobj, 'dgemm_tn', lib.PAMD_dgemm_tn, _ptr(Xv), _c.c_int(ldx), _ptr(Yv), _c.c_int(ldx),
                      _ptr(parts[ik]), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao), _c.c_long(nb * rpad), _c.c_int(0 | 2),
                      _c.c_int(nsplit), st)
        for ik, k in enumerate(members):
            _call(dfobj, 'reduce_splits', lib.PAMD_reduce_splits, _ptr(parts[ik]), _c.c_int(nsplit), _c.c_int(nao),
                  _c.c_int(nao), _ptr(vk[k]), _c.c_int(nao), _c.c_int(0), st)
    if sym:
        vk = vk + vk.transpose(1, 2)
    return vk


def get_j(dfobj, dm, hermi=0, direct_scf_tol=1e-13):
    """Integral-direct J without the 3-index tensor (pyscf/df/df_jk.py:415-506): pass 1
    gamma_Q = sum_pq (pq|Q) D_pq over freshly generated AO-row slabs (:473-478), rho = j2c^-1 gamma by
    cho_solve on the host (:481-484), pass 2 J_pq = sum_Q (pq|Q) rho_Q (:493-502).  With several ranks the
    slabs are dealt round-robin and gamma / J are all-reduced."""
    import scipy.linalg
    torch = _torch()
    lib = _lib_mod.load_library()
    dms = np.asarray(dm)
    shape = dms.shape
    nao = shape[-1]
    dms = np.ascontiguousarray(dms.reshape(-1, nao, nao), dtype=np.float64)
    nset = len(dms)
    eng, low = dfobj._direct_engine()
    dev = eng.device
    naux = eng.aux.nao
    npair = nao * (nao + 1) // 2
    st = _stream()
    dms_dev = torch.from_numpy(dms).to(dev)
    dmtril = torch.empty((nset, npair), dtype=torch.float64, device=dev)
    for s0 in range(0, nset, 4):
        ns = min(4, nset - s0)
        _call(dfobj, 'pack_dm_tril', lib.PAMD_pack_dm_tril, _ptr(dms_dev[s0:s0 + ns]), _c.c_int(ns), _c.c_int(nao),
              _ptr(dmtril[s0:s0 + ns]), st)
    slabs = dfobj._direct_slabs(eng)
    rank, world = dfobj.rank, dfobj.world_size
    bufrows = max(eng.slab_rows(a, b)[1] - eng.slab_rows(a, b)[0] for a, b in slabs)
    T = dfobj._direct_buffer(bufrows, naux, dev)
    gamma = torch.zeros((nset, naux), dtype=torch.float64, device=dev)
    for i, (sh0, sh1) in enumerate(slabs):
        if i % world != rank:
            continue
        r0, r1 = eng.slab_rows(sh0, sh1)
        Tv = eng.int3c2e_slab(sh0, sh1, out=T)
        nchunk = lib.PAMD_vj_direct_pass1_worksize(_c.c_long(r1 - r0), _c.c_int(naux)) // naux
        part = torch.empty((nchunk, naux), dtype=torch.float64, device=dev)
        for s in range(nset):
            _call(dfobj, 'vj_direct_pass1', lib.PAMD_vj_direct_pass1, _ptr(Tv), _c.c_long(naux), _c.c_long(r1 - r0),
                  _c.c_int(naux), _ptr(dmtril[s, r0:r1]), _ptr(part), st)
            gamma[s] += part.sum(dim=0)
    _allreduce(dfobj, [gamma])
    rho_h = scipy.linalg.cho_solve((low, True), gamma.cpu().numpy().T).T
    rho = torch.from_numpy(np.ascontiguousarray(rho_h)).to(dev)
    vjtril = torch.zeros((nset, npair), dtype=torch.float64, device=dev)
    for i, (sh0, sh1) in enumerate(slabs):
        if i % world != rank:
            continue
        r0, r1 = eng.slab_rows(sh0, sh1)
        Tv = eng.int3c2e_slab(sh0, sh1, out=T)
        for s in range(nset):
            _call(dfobj, 'vj_direct_pass2', lib.PAMD_vj_direct_pass2, _ptr(Tv), _c.c_long(naux), _c.c_long(r1 - r0),
                  _c.c_int(naux), _ptr(rho[s]), _ptr(vjtril[s, r0:r1]), st)
    _allreduce(dfobj, [vjtril])
    return _lib_mod.unpack_tril(vjtril.cpu().numpy(), 1).reshape(shape)


def _dm_orbital_mismatch(dms_dev, orb_list, nao):
    """0-dim device tensor max_s |D_s v - orb_s (orb_s^T v)| / max(1, |D_s v|) for one fixed pseudo-random vector: asynchronous
    (no host sync here); ~0 when every D_s equals orb_s orb_s^T."""
    torch = _torch()
    gen = torch.Generator(device='cpu').manual_seed(20240601)
    v = torch.rand(nao, dtype=torch.float64, generator=gen).to(dms_dev.device) - 0.5
    worst = torch.zeros((), dtype=torch.float64, device=dms_dev.device)
    for s_, (orb_s, _np, _ld) in enumerate(orb_list):
        c_s = orb_s[:nao]
        dv = dms_dev[s_] @ v
        worst = torch.maximum(worst, (dv - c_s @ (c_s.T @ v)).abs().max() / dv.abs().max().clamp_min(1.0))
    return worst


def _dm_matches_orbitals(dms_dev, orb_list, nao):
    """True when every D_s equals orb_s orb_s^T (to 1e-10 relative): probe with one fixed pseudo-random vector."""
    import os
    torch = _torch()
    full = os.environ.get('PAMD_DEBUG_CHECK_DM', '0') not in ('', '0')
    gen = torch.Generator(device='cpu').manual_seed(20240601)
    v = torch.rand(nao, dtype=torch.float64, generator=gen).to(dms_dev.device) - 0.5
    worst = torch.zeros((), dtype=torch.float64, device=dms_dev.device)
    for s_, (orb_s, _np, _ld) in enumerate(orb_list):
        c_s = orb_s[:nao]
        if full:
            err = (dms_dev[s_] - c_s @ c_s.T).abs().max() / dms_dev[s_].abs().max().clamp_min(1.0)
        else:
            dv = dms_dev[s_] @ v
            err = (dv - c_s @ (c_s.T @ v)).abs().max() / dv.abs().max().clamp_min(1.0)
        worst = torch.maximum(worst, err)
    return float(worst) <= 1e-10


class _HostDM:
    """Density matrices still on the HOST, uploaded on first use (r05).  With J taken from the orbitals (fused first pass) no kernel
    of the MO branch reads the matrix: the 8 nao^2-byte upload of a pageable caller array per density (2.5-3 ms at nao 1856) then
    never happens - the host API `with_df.get_jk(dm)` costs what the device-resident step costs plus the download."""

    def __init__(self, dms, device):
        self._host, self.device, self.shape, self._t = dms, device, dms.shape, None

    def tensor(self):
        if self._t is None:
            self._t = _torch().from_numpy(self._host).to(self.device)
        return self._t


def _dm_tensor(d):
    return d.tensor() if isinstance(d, _HostDM) else d


def _host_dm_mismatch(dms, blocks, own_tag=False):
    """max_s |D_s v - C_s (C_s^T v)| / max(1, |D_s v|) on the host for one fixed pseudo-random vector (the probe of the C handle's
    binding as well, pyscf_amd/df/native.py).  r06: the FULL matrix for this package's own make_rdm1 tag too - the
    every-16th-row probe of r05 missed sparse in-place edits (dm[1, 2] += h; dm[2, 1] += h of a finite-difference Fock: 1e-15
    against 8e-5 on the full matrix) and J then came silently from the orbitals.  One 8 nao^2-byte read per density with the
    BLAS pool bounded (lib.bounded_matvec); here it runs beside the queued kernels."""
    return _lib_mod.dm_orbital_mismatch(dms, blocks)


def get_jk_device(dfobj, dms_dev, orb_list=None, with_j=True, with_k=True, dm_from_orbitals=None):
    """Device-resident J/K build: inputs and outputs stay in HBM.
      dms_dev   (nset, nao, nao) f64 CUDA tensor
      orb_list  None (general-DM branch) or [(orb_dev, nocc_pad, ldo)] from `pad_orbitals`
      dm_from_orbitals  True: the caller built dms_dev[s] = orb_s orb_s^T (SCF densities from make_rdm1) - the first J pass may
                come out of the half transform's epilogue unchecked; False: never; None: decide by a cheap probe
    Returns (vjtril_dev (nset, nao_pair) | None, vk_dev (nset, nao, nao) | None), already summed
    over the aux-index shards of all ranks."""
    lib = _lib_mod.load_library()
    nset, nao = dms_dev.shape[0], dms_dev.shape[-1]
    vjtril = vk_dev = None
    outs = []
    torch = _torch()
    overlap = with_j and with_k and orb_list is not None and getattr(dfobj, 'overlap_jk', False)
    if with_j and not overlap:
        vjtril = _vj(dfobj, lib, _dm_tensor(dms_dev), nset, nao)
        outs.append(vjtril)
    if with_k:
        if orb_list is not None:
            if overlap:
                # J is HBM-bound, the K SYRK is FP64-MFMA-bound and leaves register-file room: run J
                # on a second HIP stream, released once the first half transform has been queued
                # J pass 1 behind the SYRK of the first K block, pass 2 behind the SYRK of the second one
                side = dfobj._side_stream()
                holder = {}

                sq = dfobj.square_image() if hasattr(dfobj, 'square_image') else None
                fused = (len(orb_list) == nset and all(o[1] > 0 for o in orb_list) and
                         getattr(dfobj, 'fuse_j_pass1', True))
                if fused and dm_from_orbitals is not True:
                    # the epilogue sum is the first J pass only for D_s = orb_s orb_s^T (what make_rdm1 tags; the reference
                    # itself trusts the tag for K, "#TODO: test whether dm.mo_coeff matching dm", df_jk.py:340).  Callers
                    # that built D from these orbitals say so (dm_from_orbitals=True: no check, no host sync, no library
                    # GEMM in the hot loop); otherwise a random-vector probe D v = C (C^T v) decides - two GEMVs and one
                    # scalar read-back instead of the nao^2 nocc product; PAMD_DEBUG_CHECK_DM=1 restores the full comparison.
                    fused = dm_from_orbitals is None and _dm_matches_orbitals(_dm_tensor(dms_dev), orb_list, nao)
                dfobj._last_fused = bool(fused)
                if fused:
                    # pass 1 comes out of the half transform (PAMD_nr_e2_square); pass 2 of each K block follows either on the
                    # side stream behind that block's SYRK (plain SYRK grid) or in line before it (re-tiled + balanced SYRK):
                    # which one is faster depends on the shape (config 3: overlapped 109.5 vs 111.7 ms; taxol on one GPU:
                    # 351 vs 332 ms) - `DF.j2_policy = 'auto'` times both once per shape on tensors of 4 GB and more
                    naux_l, npair_l = dfobj.tensor_shape()
                    square_layout = _square_rows(dfobj) is not None

                    def run_fused(serial):
                        rho_f = torch.zeros((nset, naux_l), dtype=torch.float64, device=dms_dev.device)
                        vj_f = torch.zeros((nset, npair_l), dtype=torch.float64, device=dms_dev.device)
                        if serial == 'fused':                       # r05: second pass inside the SYRK kernel (PAMD_syrk_jfused)
                            vk = _vk_mo(dfobj, lib, orb_list, nao, after_e2=None, fuse_j=rho_f, j_fused=vj_f)
                            return vj_f, vk

                        def pass2_block(b0, nb, iset):
                            if serial:
                                _vj_pass2_rows(dfobj, lib, b0, nb, nao, _ptr(rho_f[iset, b0:]), 1, _ptr(vj_f[iset]), _stream())
                                return
                            ev = torch.cuda.Event()
                            ev.record()
                            side.wait_event(ev)
                            with torch.cuda.stream(side):
                                _vj_pass2_rows(dfobj, lib, b0, nb, nao, _ptr(rho_f[iset, b0:]), 1, _ptr(vj_f[iset]),
                                               _c.c_void_p(side.cuda_stream))
                        vk = _vk_mo(dfobj, lib, orb_list, nao, after_e2=pass2_block, fuse_j=rho_f, j_corun=not serial)
                        return vj_f, vk

                    policy = getattr(dfobj, 'j2_policy', 'auto')
                    if policy == 'fused' and square_layout:
                        policy = 'overlap'                 # (the in-SYRK pass streams PACKED rows: a schedule of the packed layout only)
                    if policy == 'auto':
                        key = (nset, nao, tuple(o[1] for o in orb_list), naux_l,
                               0 if sq is None else sq.shape[0])
                        cache = dfobj.__dict__.setdefault('_j2_policy_cache', {})
                        policy = cache.get(key)
                        if policy is None and naux_l * npair_l * 8 < getattr(dfobj, 'j2_tune_min_bytes', 4 << 30):
                            policy = cache[key] = 'overlap'
                        cands = ('overlap', 'serial') + (('fused',) if getattr(dfobj, 'j2_try_fused', False) and not square_layout else ())

                        def decide(times):
                            order = ('overlap', 'serial', 'fused')[:len(times)]          # the library's rule (PAMD_j2_schedule_pick: the C
                            ms = (_c.c_double * 3)(*[times[n] for n in order])             # handle's trials end in the same function)
                            pol = order[lib.PAMD_j2_schedule_pick(ms, _c.c_int(len(order)))]
                            cache[key] = pol
                            dfobj._j2_policy_times = dict(times, chosen=pol)
                            return pol
                        if policy is None and getattr(dfobj, 'j2_tune', 'lazy') == 'eager':
                            # trial builds before the first answer (bench.py: the schedule is settled before the timed region)
                            timer, dfobj.kernel_timer = getattr(dfobj, 'kernel_timer', None), None
                            run_fused(False)                                     # priming: lazy images, workspaces
                            times = {}
                            for name in cands:
                                # r06: the BEST of three runs per candidate - a single run now and then carries a 20 ms hiccup
                                # (profiles/r06/bench_h2o32_1gpu_default_final.json, first version: overlap timed once at 135.9 ms,
                                # 'serial' chosen, the whole bench line 5 ms slower than the schedule it should have run)
                                best_t = None
                                for _rep in range(3):
                                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                    torch.cuda.current_stream().wait_stream(side)
                                    e0.record()
                                    run_fused('fused' if name == 'fused' else name == 'serial')
                                    torch.cuda.current_stream().wait_stream(side)
                                    e1.record()
                                    e1.synchronize()
                                    t_ = e0.elapsed_time(e1)
                                    best_t = t_ if best_t is None else min(best_t, t_)
                                times[name] = best_t
                            policy = decide(times)
                            dfobj.kernel_timer = timer
                        elif policy is None:
                            # r06, the default ('lazy'): no trial builds - the caller's OWN calls are the trials (as in the C handle).
                            # The first call of a shape primes (images, work space) on 'overlap'; each following call runs the
                            # candidate with the fewest samples between two events, read back at the NEXT call (finished long
                            # before); after `j2_lazy_reps` samples of each the best is kept.  Every schedule returns the same J
                            # and K, so an SCF pays nothing for the tuning (eager: ~1 s at config 3, ~3 s at taxol, in cycle 1).
                            st = dfobj.__dict__.setdefault('_j2_lazy', {}).setdefault(
                                key, {'times': {n: [] for n in cands}, 'pending': None, 'calls': 0})
                            if st['pending'] is not None:
                                name, e0, e1 = st['pending']
                                e1.synchronize()
                                st['times'][name].append(e0.elapsed_time(e1))
                                st['pending'] = None
                            reps = getattr(dfobj, 'j2_lazy_reps', 2)
                            todo = [n for n in cands if len(st['times'][n]) < reps]
                            st['calls'] += 1
                            if st['calls'] == 1:
                                policy = 'overlap'
                            elif not todo:
                                policy = decide({n: min(t) for n, t in st['times'].items()})
                                del dfobj._j2_lazy[key]
                            else:
                                policy = min(todo, key=lambda n: len(st['times'][n]))
                                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                torch.cuda.current_stream().wait_stream(side)
                                e0.record()
                                holder['vj'], vk_dev = run_fused('fused' if policy == 'fused' else policy == 'serial')
                                torch.cuda.current_stream().wait_stream(side)
                                e1.record()
                                st['pending'] = (policy, e0, e1)
                    if 'vj' not in holder:
                        holder['vj'], vk_dev = run_fused('fused' if policy == 'fused' else policy == 'serial')

                def launch_j(*_a):
                    if 'vj' in holder:
                        return
                    ev = torch.cuda.Event()
                    ev.record()
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        if 'rho' not in holder:
                            holder['rho'] = _vj_pass1(dfobj, lib, _dm_tensor(dms_dev), nset, nao)
                            if getattr(dfobj, 'overlap_split', True):
                                return
                        holder['vj'] = _vj_pass2(dfobj, lib, holder['rho'], nset)
                if not fused:
                    vk_dev = _vk_mo(dfobj, lib, orb_list, nao, after_e2=launch_j)
                while 'vj' not in holder:       # fewer than two K blocks were queued (or none: empty shard, nocc = 0)
                    launch_j()
                vjtril = holder['vj']
                torch.cuda.current_stream().wait_stream(side)
                vjtril.record_stream(torch.cuda.current_stream())
                outs.append(vjtril)
            else:
                vk_dev = _vk_mo(dfobj, lib, orb_list, nao)
        else:
            vk_dev = _vk_general(dfobj, lib, _dm_tensor(dms_dev), nset, nao)
        outs.append(vk_dev)
    if (_comm.active(dfobj.world_size) and getattr(dfobj, '_shard_override', None) is None and orb_list is not None and
            vjtril is not None and vk_dev is not None and getattr(dfobj, 'packed_allreduce', True)):
        _allreduce_jk_packed(dfobj, lib, vjtril, vk_dev)               # MO branch: K is symmetric
    else:
        _allreduce(dfobj, outs)
    return vjtril, vk_dev


def get_jk(dfobj, dm, hermi=0, with_j=True, with_k=True, direct_scf_tol=1e-13):
    """Same contract as ``pyscf.df.df_jk.get_jk`` (df_jk.py:280): returns (vj, vk) shaped like dm."""
    assert with_j or with_k
    torch = _torch()
    if (not with_k and not dfobj.has_tensor() and dfobj._cderi is None and
            not getattr(dfobj, 'incore_anyway', False)):
        # 3-index tensor not initialised: integral-direct J (df_jk.py:282-285)
        return get_j(dfobj, dm, hermi, direct_scf_tol), None
    if not dfobj.has_tensor() and getattr(dfobj, '_native', None) is None:
        dfobj.build()
    if getattr(dfobj, '_native', None) is not None:
        # the tensor did not fit the device: the C handle holds it (HBM + page-locked host rows, DF.build) and answers - through
        # DF.get_jk, which sums a rank's partial result over the ranks
        return dfobj.get_jk(dm, hermi, with_j, with_k, direct_scf_tol)
    dms = np.asarray(dm)
    if np.iscomplexobj(dms):
        # real/imag split, as _DFHF.get_jk does for complex DMs (df_jk.py:160-171)
        vjr, vkr = get_jk(dfobj, dms.real, 0, with_j, with_k, direct_scf_tol)
        vji, vki = get_jk(dfobj, dms.imag, 0, with_j, with_k, direct_scf_tol)
        return (vjr + 1j * vji if with_j else None), (vkr + 1j * vki if with_k else None)
    dm_shape = dms.shape
    nao = dm_shape[-1]
    dms = np.ascontiguousarray(dms.reshape(-1, nao, nao), dtype=np.float64)
    nset = dms.shape[0]
    dev = dfobj.tensor_device()
    dms_dev = _HostDM(dms, dev)                 # uploaded only if a kernel reads the matrix (not on the fused MO branch)
    # r06: where a host-API call spends its host time (bench.py `host_api_breakdown_ms`): DF.host_timing = []
    # collects one dict per call - prepare (orbital blocks, padding, upload), queue (kernel launches), probe (tag check on the
    # host beside the running kernels), download (wait for the device + the copy into page-locked arrays)
    import time as _time
    _ht = getattr(dfobj, 'host_timing', None)
    _t0 = _time.perf_counter()
    lowrank = getattr(dm, 'lowrank', None)
    if with_k and lowrank is not None and getattr(dfobj, 'lowrank_exchange', True):
        # factorised trial densities (tag: lowrank = (lefts, rights, sym), D_k = L_k R_k^T [+ h.c.]): J from the full
        # matrices as usual, K from the factors
        lib = _lib_mod.load_library()
        lefts, rights, sym = lowrank
        assert len(lefts) == nset == len(rights)
        vjtril = _vj(dfobj, lib, dms_dev.tensor(), nset, nao) if with_j else None
        vk_dev = _vk_lowrank(dfobj, lib, lefts, rights, sym, nao)
        _allreduce(dfobj, [t for t in (vjtril, vk_dev) if t is not None])
        return _to_host(dfobj, vjtril, vk_dev, nset, nao, dm_shape, with_j, with_k)
    orb_list = None
    mo_coeff = getattr(dm, 'mo_coeff', None)
    if with_k and mo_coeff is not None:
        mo_coeff = np.asarray(mo_coeff)
        mo_occ = np.asarray(dm.mo_occ)
        nmo = mo_occ.shape[-1]
        mo_coeff = mo_coeff.reshape(-1, nao, nmo)
        mo_occ = mo_occ.reshape(-1, nmo)
        if mo_occ.shape[0] * 2 == nset:      # ROHF-style DM (df_jk.py:346-351)
            mo_coeff = np.vstack((mo_coeff, mo_coeff))
            mo_occa = np.array(mo_occ > 0, dtype=np.double)
            mo_occb = np.array(mo_occ == 2, dtype=np.double)
            mo_occ = np.vstack((mo_occa, mo_occb))
        host_blocks = [mo_coeff[k][:, mo_occ[k] > 0] * np.sqrt(mo_occ[k][mo_occ[k] > 0]) for k in range(nset)]
        orb_list = [pad_orbitals(b, dev) for b in host_blocks]
    neg_sets = None
    if with_k and orb_list is None and hermi == 1 and getattr(dfobj, 'factorize_hermitian_dm', True):
        # No orbitals came with the density (initial guesses, user-built DMs): the reference then pays 4 naux nao^3
        # (df_jk.py:382-407).  A symmetric D factorises on the device as D = C+ C+^T - C- C-^T (eigenvectors scaled by
        # sqrt|w|), K is linear in D, so the MO kernels apply at 3 naux nao^2 (r+ + r-) - for the rank of a minimal-basis
        # guess an order of magnitude less.
        pos, neg = [], []
        for k in range(nset):
            dk = dms_dev.tensor()[k]
            w, v = torch.linalg.eigh((dk + dk.T) * .5)
            thr = 1e-13 * max(float(w.abs().max()), 1e-300)
            cp = (v[:, w > thr] * w[w > thr].sqrt()).cpu().numpy()
            cn = (v[:, w < -thr] * (-w[w < -thr]).sqrt()).cpu().numpy()
            pos.append(pad_orbitals(cp, dev))
            neg.append(pad_orbitals(cn, dev) if cn.shape[1] else None)
        orb_list = pos
        if any(n is not None for n in neg):
            neg_sets = neg
    promise = None
    check = None
    if orb_list is not None and mo_coeff is not None and with_j and with_k:
        # is D_s = orb_s orb_s^T?  make_rdm1 of this package says so in the tag.  For a foreign tag (stock PySCF's
        # lib.tag_array(dm, mo_coeff=, mo_occ=); the reference itself trusts it for K, df_jk.py:340) the fused first J pass runs
        # OPTIMISTICALLY and a device-side probe  max |D v - C (C^T v)|  travels back with the results: nothing waits for it, and
        # in the rare case of a tag that does not match its matrix J is redone from the matrix by the two-pass kernels below.
        # r04 (ADVICE): the package's own tag is probed as well - a tagged array edited in place (dm *= .5, dm[...] += x) keeps its
        # attributes; only the internal device loop (get_jk_device, densities it built itself) runs unchecked
        # r05: the probe runs on the HOST, after the kernels have been queued and while they run (this thread is free until the
        # download) - the matrix is no longer uploaded for it
        promise = True
        check = lambda: _host_dm_mismatch(dms, host_blocks, bool(getattr(dm, 'dm_from_orbitals', False)))
    elif neg_sets is not None or orb_list is not None:
        promise = False if neg_sets is not None else None
    _t1 = _time.perf_counter()
    vjtril, vk_dev = get_jk_device(dfobj, dms_dev, orb_list, with_j, with_k, dm_from_orbitals=promise)
    _t2 = _time.perf_counter()
    if neg_sets is not None:
        lib = _lib_mod.load_library()
        idx = [k for k in range(nset) if neg_sets[k] is not None]
        vk_neg = _vk_mo(dfobj, lib, [neg_sets[k] for k in idx], nao)
        _allreduce(dfobj, [vk_neg])
        for j, k in enumerate(idx):
            vk_dev[k] -= vk_neg[j]
    _t3 = _time.perf_counter()
    mismatch = check() if check is not None else 0.0          # (host arithmetic beside the queued kernels)
    _t4 = _time.perf_counter()
    vj, vk = _to_host(dfobj, vjtril, vk_dev, nset, nao, dm_shape, with_j, with_k)
    if _ht is not None:
        _t5 = _time.perf_counter()
        _ht.append({'prepare': round((_t1 - _t0) * 1e3, 3), 'queue': round((_t2 - _t1) * 1e3, 3), 'probe': round((_t4 - _t3) * 1e3, 3),
                    'wait_and_download': round((_t5 - _t4) * 1e3, 3)})
    if mismatch > 1e-10:
        # the tag did not describe the matrix: J from the matrix itself (the K of the MO branch follows the tag, as in the reference)
        lib = _lib_mod.load_library()
        vjtril = _vj(dfobj, lib, dms_dev.tensor(), nset, nao)
        _allreduce(dfobj, [vjtril])
        vj = _to_host(dfobj, vjtril, None, nset, nao, dm_shape, True, False)[0]
    return vj, vk


def _to_host(dfobj, vjtril, vk_dev, nset, nao, dm_shape, with_j, with_k):
    """Results to the host: J is unpacked on the device (PAMD_unpack_tril), J and K leave through one pinned staging
    buffer (a pageable download runs at a fraction of the PCIe rate; the host-side unpack cost another 10 ms at nao 1856)."""
    torch = _torch()
    dev = (vjtril if with_j else vk_dev).device
    parts = []
    if with_j:
        vj_dev = torch.empty((nset, nao, nao), dtype=torch.float64, device=dev)
        lib = _lib_mod.load_library()
        _call(dfobj, 'unpack_tril', lib.PAMD_unpack_tril, _ptr(vjtril), _c.c_long(vjtril.shape[1]), _c.c_int(nset),
              _c.c_int(nao), _ptr(vj_dev), _c.c_int(nao), _c.c_int(nao), _stream())
        parts.append(vj_dev)
    if with_k:
        parts.append(vk_dev)
    host = _download(dfobj, parts)
    vj = host.pop(0).reshape(dm_shape) if with_j else None
    vk = host.pop(0).reshape(dm_shape) if with_k else None
    return vj, vk


def _download(dfobj, tensors):
    """Device tensors -> fresh numpy arrays through a persistent pinned buffer (one async copy per tensor, one sync)."""
    return _lib_mod.download(dfobj, tensors)
