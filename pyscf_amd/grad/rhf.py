"""Analytic nuclear gradients of density-fitted RHF / UHF on the MI355X path.

Mirror of ``pyscf/grad/rhf.py`` (``grad_elec`` :36-88, ``grad_nuc`` :148-166, ``hcore_generator`` :91-146,
``make_rdm1e`` :196-205) with the J/K part of ``pyscf/df/grad/rhf.py`` (``get_jk`` :44-260,
``auxbasis_response``), and ``pyscf/grad/uhf.py`` / ``pyscf/df/grad/uhf.py`` for two spin blocks.

The reference materialises derivative-integral tensors (int3c2e_ip1, int3c2e_ip2, int2c2e_ip1, int1e_ipovlp,
int1e_ipkin, int1e_ipnuc, int1e_iprinv) and contracts them on the host.  Here the DF energy is differentiated
as a function of the raw integrals A = (Q|pq) and the metric M = (P|Q),

    E_2e = 1/2 g^T M^-1 g - k/2 sum_s sum_PQ M^-1_PQ Tr[D_s A_P D_s A_Q],     g_Q = sum_pq A_Q,pq D_pq
    dE   = sum_{Q,pq} Z_Q,pq dA_Q,pq + sum_PQ Y_PQ dM_PQ
    Z_Q  = sum_L (L^-T)_QL [ rho_L D - k sum_s C_s (C_s^T B_L C_s) C_s^T ]
    Y    = L^-T [ -1/2 rho rho^T + k/2 sum_s sum_ij y^s_L,ij y^s_L',ij ] L^-1,  y^s_L = C_s^T B_L C_s

(B = cderi, rho = B d, M = L L^T), and one generate-and-contract kernel family (``PAMD_int3c2e_grad_class``)
dots every shell triple's derivative block with the matching block of Z; the same kernel with the aux shells
replaced by (P, unit s) pairs gives the metric term and with point charges the nuclear-attraction term.
The derivative on the third centre comes from translational invariance inside the kernel, so
``auxbasis_response=False`` (df/grad/rhf.py:133-138) is a kernel flag.
"""
import ctypes as _c

import numpy as np

from .. import lib as _lib_mod
from ..df import df_jk
from ..df.incore import _decompose_j2c
from ..gto.moleintor import _AuxClass, _Shells, _dev, c2s_matrix, get_engine

NREP = 64          # replicated accumulators: spreads the FP64 atomics of the contraction kernel



def _need_torch_df(dfobj):
    """The gradient kernels work on the torch-resident tensor of pyscf_amd.df.DF.  `mf.density_fit(devices=...)` / PAMD_DEVICES put
    the host-array handle object (NativeDF) in its place, which has no `_cderi_dev`: say so instead of an AttributeError
    (ADVICE r04)."""
    from ..df.native import NativeDF
    if isinstance(dfobj, NativeDF):
        raise NotImplementedError('analytic DF gradients need the torch-resident pyscf_amd.df.DF; this SCF object was routed to the '
                                  'host-array handle (density_fit(devices=...) or PAMD_DEVICES): rebuild it with mf.density_fit() '
                                  '(one process per GPU under torch.distributed for several devices)')


def grad_nuc(mol, atmlst=None):
    """pyscf/grad/rhf.py:148-166."""
    z = mol.atom_charges().astype(float)
    r = mol.atom_coords()
    natm = len(z)
    g = np.zeros((natm, 3))
    for i in range(natm):
        for j in range(natm):
            if i != j and z[i] != 0 and z[j] != 0:      # ghost atoms may sit on top of real ones (cf. energy_nuc)
                d = r[i] - r[j]
                g[i] -= z[i] * z[j] * d / np.linalg.norm(d) ** 3
    return g if atmlst is None else g[atmlst]


def _dbg(msg):
    import os
    if os.environ.get('PAMD_DEBUG_GRAD'):
        import torch
        torch.cuda.synchronize()
        print('[grad]', msg, flush=True)


def _ptr(t):
    return _c.c_void_p(t.data_ptr())


def _pack_tril_dev(m):
    """(..., nao, nao) device -> (..., nao_pair) lower-triangular packed, p(p+1)/2+q."""
    import torch
    nao = m.shape[-1]
    ti, tj = np.tril_indices(nao)
    idx = torch.from_numpy(ti * nao + tj).to(m.device)
    return m.reshape(*m.shape[:-2], nao * nao)[..., idx]


def two_particle_densities(dfobj, dm_tot, occ_blocks, kscale, mh, jscale=1.0):
    """W[L][pq] (local aux rows x nao_pair), the local rows of M (cderi = M (Q|pq)) and Y[P][Q] (naux, naux) of the module
    docstring, on the device - Z_T[pq][Q] = sum_L W[L][pq] M[L][Q] is formed slab by slab by the caller (z_slab); the Coulomb
    part scaled by jscale (0 for the exchange-only terms of range-separated hybrids).

    occ_blocks: [(C (nao, nocc) with D_s = C C^T, weight)], e.g. RHF [(C_occ, 2)], UHF [(Ca, 1), (Cb, 1)].
    mh: (rows of cderi, naux) host matrix with cderi = mh (Q|pq), i.e. L^-1 of the metric's Cholesky factor or, for a
    linearly dependent metric, (V / sqrt(w))^T - then M^-1 of the docstring is the pseudo-inverse mh^T mh held fixed, as
    in the reference's 'ED' solver (df/grad/rhf.py:434-443)."""
    import torch
    so = _lib_mod.load_library()
    cderi = dfobj._cderi_dev
    naux, npair = cderi.shape
    dev = cderi.device
    nao = dm_tot.shape[0]
    f64 = torch.float64
    st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_dev = torch.from_numpy(np.ascontiguousarray(dm_tot)).to(dev)
    dtril = _pack_tril_dev(d_dev)
    dsum = dtril * 2
    diag = torch.from_numpy(np.arange(nao) * (np.arange(nao) + 1) // 2 + np.arange(nao)).to(dev)
    dsum[diag] *= .5
    # aux-sharded tensor (world > 1): this rank holds rows [l0, l1) of cderi.  rho, W, y are row-local; ytil needs the
    # rho and y of ALL rows (gathered by an all-reduce of zero-padded buffers: works on RCCL and gloo alike), and the
    # (Z, Y) built below are this rank's PARTIAL sums over its rows - the gradient is linear in them, so every rank
    # contracts its partials with the derivative integrals and the (natm, 3) results are all-reduced (_grad_2e).
    world = dfobj.world_size if getattr(dfobj, '_shard_override', None) is None else 1
    naux_all = mh.shape[0]
    l0, l1 = dfobj.shard_range(naux_all, dfobj.rank, dfobj.world_size) if world > 1 else (0, naux)
    assert l1 - l0 == naux

    def gather_rows(x):
        """[naux_local, ...] -> [naux_all, ...] over the ranks"""
        if world == 1:
            return x
        import torch.distributed as dist
        full = torch.zeros((naux_all,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        full[l0:l1] = x
        dist.all_reduce(full, group=dfobj.group)
        return full
    rho = cderi @ dsum * jscale                                  # rho_L = sum_pq B_L,pq D_pq (full square)
    W = rho[:, None] * dtril[None, :]                            # [L][pq]
    rho_all = gather_rows(rho)
    ytil = -0.5 / jscale * torch.outer(rho, rho_all) if jscale else torch.zeros((naux, naux_all), dtype=f64, device=dev)
    ldx = (nao + 15) // 16 * 16
    for c, wgt in occ_blocks:
        nocc = c.shape[1]
        if nocc == 0 or kscale == 0:
            continue
        orb, nocc_pad, ldo = df_jk.pad_orbitals(np.asarray(c, dtype=np.float64), dev)
        c_dev = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float64)).to(dev)
        blk = max(1, min(naux, int((2 << 30) // (nocc_pad * ldx * 8)), int((4 << 30) // (nao * nao * 8))))
        ys = torch.empty((naux, nocc * nocc), dtype=f64, device=dev)
        for b0 in range(0, naux, blk):
            nb = min(blk, naux - b0)
            X = torch.zeros((nb, nocc_pad, ldx), dtype=f64, device=dev)
            df_jk._call(dfobj, 'e2_symm', so.PAMD_nr_e2_symm, _ptr(cderi[b0:b0 + nb]), _c.c_long(npair), _c.c_int(nb),
                        _c.c_int(nao), _ptr(orb), _c.c_int(ldo), _c.c_int(orb.shape[0]), _c.c_int(nocc_pad), _ptr(X),
                        _c.c_int(ldx), _c.c_void_p(0), _c.c_void_p(0), st)
            y = torch.matmul(X[:, :nocc, :nao], c_dev)           # y_L,ij = (C^T B_L C)_ij
            ys[b0:b0 + nb] = y.reshape(nb, -1)
            cyc = torch.matmul(c_dev, torch.matmul(y, c_dev.T))  # C y_L C^T
            W[b0:b0 + nb] -= kscale * wgt * _pack_tril_dev(cyc)
        ytil += 0.5 * kscale * wgt * (ys @ gather_rows(ys).T)
    linv = torch.from_numpy(np.ascontiguousarray(mh)).to(dev)    # all rows: [naux_all][nq]
    linv_loc = linv[l0:l1]
    y_pq = linv_loc.T @ (ytil @ linv)
    # r04: Z_T[pq][Q] = sum_L W[L][pq] Linv[L][Q] is NOT formed here any more (it has the size of the whole tensor per rank):
    # _grad_2e forms it AO-row slab by slab from (W, linv_loc) - see there
    return W, linv_loc, y_pq


def _timed(dfobj, name, fn):
    """Run fn() between two HIP events when the object carries a kernel timer (tools/grad_bench.py: per-phase ms of the gradient)."""
    timer = getattr(dfobj, 'kernel_timer', None)
    if timer is None:
        return fn()
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    timer.records.append((name, e0, e1))
    return out


def z_slab(W, linv_loc, r0, r1, out=None, dfobj=None, work=None):
    """Z_T[pq][Q] = sum_L W[L][pq] M[L][Q] for the packed rows pq in [r0, r1), summed over the LOCAL aux rows: (r1 - r0, nq) - the
    naux^2 nao_pair flops of the gradient (pyscf/df/grad/rhf.py:117-199 streams the same contraction block by block).
    r06: the library's own FP64-MFMA TN GEMM (PAMD_dgemm_tn, 160 x 128 tiles, LDS-DMA operands) instead of
    torch.matmul = rocBLAS.  Its A operand must be addressable inside one 4 GiB window per k-range, which W[L][pq] (13.8 MB per
    aux row at config 3) is not: the slab's columns are first copied into a contiguous [k][ncol] work buffer - the slab is <= 4 GB,
    the copy moves 2 x 4 GB against 2 naux^2 ncol flops.  `work` = (wbuf, mbuf): the work buffer and M padded to a multiple of 16
    rows (zero rows), made once per gradient.
    MEASURED (profiles/r06/grad_h2o32_rhf_{hip,torch}.json, config-3 size, one box): the library GEMM runs the 6.8e13 flops in
    935.6 ms = 72.9 TF/s = 0.927 of the FP64 matrix peak, the hand-written path in 1024.4 ms = 66.6 TF/s = 0.847 (its 160 x 128
    tile kernel is the r01 DMA scheme, and it pays the slab copy).  A plain GEMM at 0.93 is what the task leaves to the library:
    DF.grad_z_gemm = 'torch' is the default, 'hip' keeps the hand-written path selectable."""
    import torch
    if work is None or getattr(dfobj, 'grad_z_gemm', 'torch') != 'hip':
        return _timed(dfobj, 'z_slab_gemm', lambda: torch.matmul(W[:, r0:r1].T, linv_loc, out=out))
    wbuf, mbuf = work
    k, ncol = W.shape[0], r1 - r0
    k16, nq = mbuf.shape
    lda = (ncol + 1) // 2 * 2
    wb = wbuf[:k16 * lda].view(k16, lda)
    if lda != ncol:
        wb[:, ncol:].zero_()
    wb[:k, :ncol].copy_(W[:, r0:r1])
    if k16 > k:
        wb[k:].zero_()
    if out is None:
        out = torch.empty((ncol, nq), dtype=torch.float64, device=W.device)
    out.zero_()                                         # the GEMM accumulates (FP64 atomics of its split-K form; one split here)
    so = _lib_mod.load_library()
    st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
    _timed(dfobj, 'z_slab_gemm', lambda: _lib_mod.check(so.PAMD_dgemm_tn(
        _ptr(wb), _c.c_int(lda), _ptr(mbuf), _c.c_int(nq), _ptr(out), _c.c_int(nq), _c.c_int(ncol), _c.c_int(nq),
        _c.c_long(k16), _c.c_int(2), _c.c_int(1), st)))
    return out


def _grad_2e(mol, dfobj, dm_tot, occ_blocks, jscale, kscale, auxbasis_response, grad):
    """grad[rep][atom][3] += sum Z dA + sum Y dM of the DF two-electron energy jscale J - kscale K carried by dfobj's
    tensor; returns the integral engine.  A long-range tensor (dfobj.omega > 0) differentiates the erf-attenuated
    integrals; a short-range one (omega < 0) the Coulomb integrals with (Z, Y) and the long-range ones with (-Z, -Y)."""
    import torch
    _need_torch_df(dfobj)
    if not dfobj.has_tensor() and getattr(dfobj, '_native', None) is None:
        dfobj.build()
    if getattr(dfobj, '_native', None) is not None:
        raise NotImplementedError('analytic gradients need the in-core tensor; this DF object holds it out of core (C handle): '
                                  'shard the auxiliary index over more ranks')
    dev = dfobj.tensor_device()
    dfobj.drop_square_image()            # W below takes the size of cderi (square layout: packed first, DF.to_packed_layout)
    sharded = dfobj.world_size > 1 and getattr(dfobj, '_shard_override', None) is None
    eng = get_engine(mol, dfobj.auxmol, dev, dfobj.omega)
    naux = eng.aux.nao
    if sharded:
        grad_total, grad = grad, torch.zeros_like(grad)
    j2c = eng.int2c2e().cpu().numpy()
    mh = _decompose_j2c((j2c + j2c.T) * .5, dfobj.lindep, getattr(dfobj, 'decompose_j2c', 'CD'), dev)[0]     # as in the tensor build
    nrow_local = dfobj._cderi_dev.shape[0]
    l0, l1 = dfobj.shard_range(mh.shape[0], dfobj.rank, dfobj.world_size) if sharded else (0, mh.shape[0])
    if l1 - l0 != nrow_local:
        raise RuntimeError('the tensor has %d rows here, the metric decomposes into %d (this rank: %d)'
                           % (nrow_local, mh.shape[0], l1 - l0))
    W, linv_loc, y_pq = _timed(dfobj, 'two_particle_densities', lambda: two_particle_densities(dfobj, dm_tot, occ_blocks, kscale, mh, jscale))
    y_pq = y_pq.contiguous()
    _dbg('W,Y built')
    nq = linv_loc.shape[1]
    ao_atom = _dev(eng.ao.atom, dev)
    aux_atom = _dev(eng.aux.atom, dev)
    aux_xyz, aux_ao0 = _dev(eng.aux.xyz, dev), _dev(eng.aux.ao0, dev)
    # r04 - the scalable form (the reference streams the same contraction block by block,
    # pyscf/df/grad/rhf.py:117-199).  Z_T[pq][Q] is formed and consumed AO-row slab by slab: a slab of packed rows [r0, r1) is
    # one GEMM over the LOCAL aux rows, W[:, r0:r1]^T M_loc -> (r1 - r0, nq); on an aux-sharded tensor the ranks' partial slabs
    # are summed onto the slab's OWNER (slabs dealt round-robin; RCCL reduce, all-reduce on gloo) and only the owner runs the
    # slab's derivative-integral kernels (pair sub-ranges by row shell, Z addressed with a row offset - the addressing of the
    # tensor build).  Per rank: its rows of W (the size of its cderi shard), one slab buffer, 1 / world of the GEMM AND of the
    # derivative-integral work - nothing of the size of the whole tensor, so config-5 shapes no longer refuse.
    slab_bytes = int(getattr(dfobj, 'grad_slab_bytes', 4 << 30))
    max_rows = max(1, slab_bytes // (nq * 8))
    slabs, sh0 = [], 0
    nsh = eng.ao.n
    while sh0 < nsh:
        sh1 = sh0 + 1
        while sh1 < nsh and eng.slab_rows(sh0, sh1 + 1)[1] - eng.slab_rows(sh0, sh1 + 1)[0] <= max_rows:
            sh1 += 1
        slabs.append((sh0, sh1))
        sh0 = sh1
    bufrows = max(eng.slab_rows(a, b)[1] - eng.slab_rows(a, b)[0] for a, b in slabs)
    zbuf = torch.empty((bufrows, nq), dtype=torch.float64, device=dev)
    dfobj._grad_slabs = len(slabs)
    # operands of the hand-written slab GEMM (z_slab): a contiguous [k16][slab columns] copy buffer + 256 doubles of slack for the
    # LDS-DMA kernels' whole-panel reads, and M with its rows padded to a multiple of 16; needs even nq and a 16-byte aligned M
    zwork = None
    k16 = (nrow_local + 15) // 16 * 16
    if getattr(dfobj, 'grad_z_gemm', 'torch') == 'hip' and nq % 2 == 0 and nrow_local > 0:
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        need = (k16 * (bufrows + 2) + k16 * nq + 512) * 8
        if need + (2 << 30) <= free:
            wbuf = torch.zeros(k16 * (bufrows + 2) + 256, dtype=torch.float64, device=dev)
            mbuf = torch.zeros(k16 * nq + 256, dtype=torch.float64, device=dev)[:k16 * nq].view(k16, nq)
            mbuf[:nrow_local].copy_(linv_loc)
            zwork = (wbuf, mbuf)
    passes = [None] if dfobj.omega >= 0 else [0.0, -dfobj.omega]
    world, rank = (dfobj.world_size, dfobj.rank) if sharded else (1, 0)
    try:
        for islab, (sa, sb) in enumerate(slabs):
            r0, r1 = eng.slab_rows(sa, sb)
            z = z_slab(W, linv_loc, r0, r1, out=zbuf[:r1 - r0], dfobj=dfobj, work=zwork)
            owner = islab % world
            if sharded:
                import torch.distributed as dist
                from ..lib import comm as _comm
                if _comm.backend_name(dfobj.group) == 'nccl':
                    dist.reduce(z, dst=dist.get_global_rank(dfobj.group, owner) if dfobj.group is not None else owner, group=dfobj.group)
                else:
                    dist.all_reduce(z, group=dfobj.group)
                if owner != rank:
                    continue
            for ip, om in enumerate(passes):
                if om is not None:
                    eng._omega_override = om
                if ip == 1:
                    z.neg_()
                # (1) sum Z dA: 3-centre derivative blocks of this slab
                def _deriv_blocks():
                    for pc in eng.pair_classes():
                        i0, i1 = pc.subrange(sa, sb)
                        if i1 <= i0:
                            continue
                        for ac in eng.aux_classes():
                            eng.grad_launch(pc, ac, z, nq, 1, eng.ao_xyz, eng.ao_ao0, ao_atom, grad, auxbasis_response, i0=i0, i1=i1,
                                            row_offset=r0)
                _timed(dfobj, 'int3c2e_grad', _deriv_blocks)
        _dbg('3c slabs done')
        # (2) sum Y dM: 2-centre metric (every rank its partial Y)
        if auxbasis_response:
            for ip, om in enumerate(passes):
                if om is not None:
                    eng._omega_override = om
                if ip == 1:
                    y_pq.neg_()
                for pc in eng.pair_classes_2c():
                    for ac in eng.aux_classes():
                        eng.grad_launch(pc, ac, y_pq, naux, 0, aux_xyz, aux_ao0, aux_atom, grad, True)
        torch.cuda.synchronize()         # W / zbuf / y_pq are released on return
    finally:
        if hasattr(eng, '_omega_override'):
            del eng._omega_override
    if sharded:
        import torch.distributed as dist
        dist.all_reduce(grad, group=dfobj.group)            # sum of the ranks' partial two-electron gradients
        grad_total += grad
    _dbg('2e done')
    return eng


def grad_elec_df(mol, dfobj, dm_tot, occ_blocks, dme, kscale=1.0, auxbasis_response=True, device=None, exchange_terms=()):
    """Electronic gradient (natm, 3) of a DF-HF-type energy with total density dm_tot, exchange from
    occ_blocks (see two_particle_densities) scaled by kscale, energy-weighted density dme.
    exchange_terms: [(range-separated DF object, scale)] further -scale K terms on other tensors (the long- or short-range
    exact exchange of range-separated hybrids, pyscf/df/grad/rks.py:84-110)."""
    import torch
    so = _lib_mod.load_library()
    _need_torch_df(dfobj)
    if not dfobj.has_tensor() and getattr(dfobj, '_native', None) is None:
        dfobj.build()
    if getattr(dfobj, '_native', None) is not None:
        raise NotImplementedError('analytic gradients need the in-core tensor; this DF object holds it out of core (C handle): '
                                  'shard the auxiliary index over more ranks')
    dev = dfobj.tensor_device()
    natm = mol.natm
    grad = torch.zeros((NREP, natm, 3), dtype=torch.float64, device=dev)
    eng = _grad_2e(mol, dfobj, dm_tot, occ_blocks, 1.0, kscale, auxbasis_response, grad)
    for rs_df, scale in exchange_terms:
        if scale != 0:
            _grad_2e(mol, rs_df, dm_tot, occ_blocks, 0.0, scale, auxbasis_response, grad)
    nao = eng.ao.nao
    ao_atom = _dev(eng.ao.atom, dev)
    # (3) nuclear attraction: point-charge "aux shells", Z[pq][C] = D_pq (the charge -Z_C sits in the coefficient)
    nuc = _Shells.__new__(_Shells)
    eta = 1e30
    nuc.l = np.zeros(natm, np.int32)
    nuc.xyz = mol.atom_coords()
    nuc.exps = [np.array([eta])] * natm
    zc = mol.atom_charges().astype(float)
    nuc.coefs = [np.array([-zi * (eta / np.pi) ** 1.5 / c2s_matrix(0)[0, 0]]) for zi in zc]
    nuc.ao0 = np.arange(natm, dtype=np.int32)
    nuc.atom = np.arange(natm, dtype=np.int32)
    nuc.n = nuc.nao = natm
    ac = _AuxClass(nuc, 0, dev)
    d_dev = torch.from_numpy(np.ascontiguousarray(dm_tot)).to(dev)
    z_nuc = _pack_tril_dev(d_dev)[:, None].expand(-1, natm).contiguous()
    eng._omega_override = 0.0
    try:
        for pc in eng.pair_classes():
            eng.grad_launch(pc, ac, z_nuc, natm, 1, eng.ao_xyz, eng.ao_ao0, ao_atom, grad, True)
    finally:
        del eng._omega_override
    _dbg('nuc done')
    g = grad.sum(dim=0)
    # (4) kinetic and overlap (energy-weighted density)
    sh = eng.ao
    prim0 = np.cumsum([0] + [len(e) for e in sh.exps])[:-1].astype(np.int32)
    nprim = np.array([len(e) for e in sh.exps], np.int32)
    st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
    w_dev = torch.from_numpy(np.ascontiguousarray(dme)).to(dev)
    tabs = [_dev(sh.l, dev), eng.ao_ao0, _dev(prim0, dev), _dev(nprim, dev), eng.ao_xyz,
            _dev(np.concatenate(sh.exps), dev), _dev(np.concatenate(sh.coefs), dev), ao_atom]      # keep alive
    _lib_mod.check(so.PAMD_int1e_grad(*[_ptr(t) for t in tabs], _c.c_int(sh.n), _c.c_int(nao), _ptr(eng.c2s),
                                      _ptr(eng.c2s_off), _ptr(d_dev), _ptr(w_dev), _ptr(g), st))
    torch.cuda.synchronize()
    _dbg('1e done')
    return g.cpu().numpy()


def rsh_exchange_terms(dfobj, omega, alpha, hyb):
    """(scale of K on the Coulomb tensor, [(range-separated DF object, scale)]): the four exchange branches of
    get_veff (dft/rks.py:108-127, df/grad/rks.py:84-110) - full-range hybrid; short-range only (erfc tensor x hyb);
    long-range only (erf tensor x alpha); both, K = hyb K_full + (alpha - hyb) K_LR."""
    if omega == 0:
        return hyb, ()
    if alpha == 0:
        return 0.0, [(dfobj.range_coulomb(-omega), hyb)]
    if hyb == 0:
        return 0.0, [(dfobj.range_coulomb(omega), alpha)]
    return hyb, [(dfobj.range_coulomb(omega), alpha - hyb)]


class Gradients:
    """``mf.nuc_grad_method()`` / ``mf.Gradients()`` of a converged DF-RHF or DF-UHF object
    (pyscf/grad/rhf.py:291-458, pyscf/df/grad/rhf.py:262-320)."""

    def __init__(self, mf):
        if getattr(mf, 'only_dfj', False):
            # density_fit(only_dfj=True): the SCF energy holds the EXACT in-core K; the reference then differentiates a DF J
            # plus an exact K (pyscf/df/grad/rhf.py:73-83).  The exact-K gradient is not built here, and differentiating a
            # DF-K energy instead would silently not be the derivative of the converged energy.
            raise NotImplementedError('nuclear gradients with density_fit(only_dfj=True) need the exact-exchange gradient '
                                      '(pyscf/df/grad/rhf.py:73-83), which is not implemented; use density_fit() with a fitted K')
        self.base = mf
        self.mol = mf.mol
        self.auxbasis_response = True
        self.de = None

    def set(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        return self

    def grad_nuc(self, mol=None, atmlst=None):
        return grad_nuc(mol or self.mol, atmlst)

    def _densities(self):
        mf = self.base
        mo_c, mo_e, mo_occ = np.asarray(mf.mo_coeff), np.asarray(mf.mo_energy), np.asarray(mf.mo_occ)
        if mo_c.ndim == 2 and np.any((mo_occ > 0) & (mo_occ < 2)):
            # ROHF (pyscf/grad/rohf.py:30-42): UHF machinery with the blocks occ > 0 / occ == 2 of the one coefficient
            # matrix; the orbitals do not diagonalise F_alpha / F_beta, so W = sum_s D_s F_s D_s, D_s = C_s C_s^T
            ca, cb = mo_c[:, mo_occ > 0], mo_c[:, mo_occ == 2]
            da, db = ca.dot(ca.T), cb.dot(cb.T)
            vhf = mf.get_veff(self.mol, np.array((da, db)))
            h1e = mf.get_hcore()
            dme = da.dot(h1e + vhf[0]).dot(da) + db.dot(h1e + vhf[1]).dot(db)
            return da + db, [(ca, 1.0), (cb, 1.0)], dme
        if mo_c.ndim == 2:
            occ = mo_occ > 0
            c = mo_c[:, occ]
            dm = 2 * c.dot(c.T)
            dme = 2 * (c * mo_e[occ]).dot(c.T)                   # make_rdm1e, grad/rhf.py:196-205
            return dm, [(c, 2.0)], dme
        blocks, dm, dme = [], 0, 0
        for s in range(2):
            occ = mo_occ[s] > 0
            c = mo_c[s][:, occ]
            blocks.append((c, 1.0))
            dm = dm + c.dot(c.T)
            dme = dme + (c * mo_e[s][occ]).dot(c.T)
        return dm, blocks, dme

    def grad_elec(self):
        mf = self.base
        if getattr(mf, 'with_df', None) is None:
            raise NotImplementedError('gradients are implemented for density-fitted SCF objects')
        dm, blocks, dme = self._densities()
        return grad_elec_df(self.mol, mf.with_df, dm, blocks, dme, 1.0, self.auxbasis_response)

    def kernel(self):
        self.de = self.grad_elec() + self.grad_nuc()
        return self.de

    grad = kernel
