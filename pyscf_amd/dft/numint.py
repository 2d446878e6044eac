"""Numerical integration of the XC potential on the MI355X.

Mirror of ``pyscf/dft/numint.py``: ``NumInt.nr_rks`` (:1074-1190), ``block_loop`` (:2887-2928),
``eval_ao`` (:51-114), ``eval_rho``/``eval_rho2`` (:116-469), ``_scale_ao`` / ``_dot_ao_ao``
(:803-874), ``rsh_and_hybrid_coeff`` (:2805-2828).  Per grid block, on the device:

    ao   = PAMD_eval_ao              (comp, grid, ldao), AO index fastest     GTOval_sph_deriv0/1
    c    = PAMD_cderi_solve (GEMM)   c[comp][i][g] = sum_mu C_occ[mu][i] ao[comp][g][mu]   (MO branch)
    rho  = PAMD_rho_from_mo / _dm    rho, grad rho
    wv   = PAMD_eval_xc              w * (vrho/2, 2 vsigma grad rho); nelec, exc accumulated
    aow  = PAMD_scale_ao             sum_c wv_c ao_c
    vmat+= PAMD_dgemm_tn (LDS-DMA)   ao0^T . aow ;  finally vmat = M + M^T

AO values are recomputed per SCF iteration block by block (exp-bound, cheap next to the two
GEMMs) instead of being stored; with several ranks the grid blocks are dealt round-robin and
vmat / nelec / exc are all-reduced.
"""
import ctypes
import os

import numpy as np

from .. import lib as _lib_mod
from ..lib import comm as _comm
from . import libxc as _xc

_c = ctypes


def _ptr(t):
    return _c.c_void_p(t.data_ptr())


def _round_up(x, m):
    return (x + m - 1) // m * m


def grid_block_size(ngrids, max_rows, world=1):
    """Rows per grid block: a multiple of 256, at most `max_rows`, and such that the number of blocks
    is a multiple of the number of ranks (blocks are dealt round-robin, block b -> rank b % world)."""
    max_rows = max(256, max_rows // 256 * 256)
    nblk = max(1, -(-ngrids // max_rows))
    if world > 1:
        nblk = max(nblk, 2 * world)
        nblk = -(-nblk // world) * world
    blk = _round_up(-(-ngrids // nblk), 256)
    return max(256, min(blk, max_rows))


def pick_nsplit(ntiles, slots=512, lo=2, hi=12):
    """Split-k factor of the vmat GEMM: fill whole rounds of the chip's workgroup slots
    (256 CUs x 2 resident 128 x 128 tiles) -- 225 tiles x 9 splits = 2025 of 2048, not 900 of 1024."""
    best, best_eff = lo, 0.0
    for ns in range(lo, hi + 1):
        wg = ntiles * ns
        eff = wg / (-(-wg // slots) * slots)
        if eff > best_eff + 1e-9:
            best, best_eff = ns, eff
    return best


def estimate_ao_image_bytes(mol, ncomp=4):
    """Upper estimate of the compact AO image the block-sparse plan will cache (dft/sparse_grid.py), before any grid exists: a
    level-3 pruned grid has ~12 500 points per atom (config 3: 11.2 k, taxol: 12.2 k measured), and a 512-point tile sees at most
    ~750 of the AO functions above the 1e-13 cutoff however large the molecule is (locality; measured means: 406 at config 3, 697
    at taxol).  Used as the XC share of the HBM budget (df.DF.xc_image_hint)."""
    nao = int(mol.nao_nr()) if hasattr(mol, 'nao_nr') else int(mol.nao)
    natm = len(mol._atm)
    return int(12500 * natm * ncomp * 8 * min(nao, 750))


class NumInt:
    """Duck-types the attributes RKS.get_veff uses (pyscf/dft/rks.py:76-131,384-404)."""
    libxc = _xc
    cutoff = 1e-13

    def __init__(self, device=None, block_bytes=6 << 30, group=None):
        self.device = device
        self.vmat_nsplit = None         # None: pick_nsplit()
        # vmat GEMM: fraction of k-tiles that must be negligible before the screened (128-row tile) kernel is preferred
        # to the unscreened 160 x 128-tile one.  Measured on (H2O)_32 cc-pVTZ: screened 115 ms (5 % skipped) vs 119-123 ms
        # unscreened wide, so screening stays on whenever it is enabled
        self.vmat_screen_min_skip = 0.0
        self.screen_cutoff = 1e-15      # |value| below which a 16 x 16 AO tile is skipped (None: dense)
        self.block_bytes = block_bytes
        self.group = group
        self._cache = {}
        self.kernel_timer = None
        # block-sparse path (dft/sparse_grid.py, csrc/xc_sparse.hip): compact AO subsets per grid tile
        self.sparse = True              # False: the dense tile-masked pipeline below (kept for comparison / tests)
        self.sparse_tile = 512          # grid points per tile (multiple of 128; 512 measured best at (H2O)_32: 30.1 ms vs 32.0 / 35.1 for 1024 / 2048)
        self.sparse_cutoff = 1e-13      # a shell is active on a tile if some value / gradient component exceeds this.  r04: 1e-13, the
                                        # reference's own threshold for the sparse contractions (numint.py:1120,2845 `cutoff = CUTOFF * 1e2`;
                                        # its AO screen itself is the 1e-15 exponent estimate): nelec, exc and vxc of config 3 unchanged to
                                        # 12 digits against 1e-14, 5 % fewer active functions (profiles/r04/xcbench_cutoff_sweep.log)
        self.sparse_chunk_points = 1 << 21  # grid points per launch group (bounds the c = ao . C workspace: 5.5 GB at config 3); r04: one
                                            # group for the whole grid - 9 launches per kernel cost 0.9 ms of tails in sub_vmat alone
        self.fuse_rho = True            # r04: GGA densities in the epilogue of the orbital product (PAMD_sub_orb_rho)
        self.vmat_sym = True            # r04: V = M + M^T on balanced blocks, lower triangle only (PAMD_sub_vmat_sym); False: the r03 kernel
        self.ao_cache = 'auto'          # keep the compact AO image in HBM across calls: True / False / 'auto' (if it fits)
        self.ao_cache_reserve = 14 << 30    # HBM left free after caching ('auto'): the plan's own work space (orbital-product chunk
                                            # buffer <= 6 GB, aow image, dense evaluation block).  r06: was a blanket 40 GB, which turned the
                                            # cache off at taxol size although 70 GB were free (VERDICT r05 Weak 4)

    # -- functional properties --------------------------------------------------------------
    def _xc_type(self, xc_code):
        return _xc.xc_type(xc_code)

    def hybrid_coeff(self, xc_code, spin=0):
        return _xc.hybrid_coeff(xc_code)

    def nlc_coeff(self, xc_code):
        return ()

    def rsh_coeff(self, xc_code):
        omega, alpha, beta = _xc.rsh_coeff(xc_code)
        if self.omega is not None and omega != 0:
            omega = float(self.omega)
        return omega, alpha, beta

    def rsh_and_hybrid_coeff(self, xc_code, spin=0):
        omega, alpha, beta = self.rsh_coeff(xc_code)
        return omega, alpha, self.hybrid_coeff(xc_code, spin)

    omega = None          # overrides the functional's range-separation parameter (KohnShamDFT.omega -> numint.omega,
                          # pyscf/dft/rks.py:445-455, numint.py:2737-2760)

    def _parse(self, xc_code):
        """(hyb, fac[NFAC]) of libxc.parse_xc with the omega override applied to the attenuated exchange."""
        hyb, fac = _xc.parse_xc(xc_code)
        if self.omega is not None and fac[_xc.F_OMEGA] != 0:
            fac = fac.copy()
            fac[_xc.F_OMEGA] = abs(float(self.omega))
        return hyb, fac

    # -- device plumbing ----------------------------------------------------------------------
    def _dev(self):
        import torch
        if self.device is not None:
            return torch.device(self.device)
        if not torch.cuda.is_available():
            raise RuntimeError('NumInt needs a HIP device (MI355X); there is no CPU fallback')
        return torch.device('cuda', torch.cuda.current_device())

    def _world(self):
        if getattr(self, '_world_override', None) is not None:      # (rank, world) without collectives: tools / tests
            return self._world_override
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_rank(self.group), dist.get_world_size(self.group)
        except ImportError:
            pass
        return 0, 1

    def _allreduce(self, tensors, world):
        """Sum the ranks' partial grid sums (vmat, nelec, exc, gradient parts): RCCL all-reduce over the grid-tile shards."""
        if getattr(self, '_world_override', None) is None:
            _comm.all_reduce(tensors, self.group, world)

    def _shell_tables(self, mol, dev):
        from ..gto.moleintor import mol_fingerprint
        key = ('shells', mol_fingerprint(mol), str(dev))       # content key: survives mol.build(atom=...) in place
        if key not in self._cache:
            from ..gto.moleintor import get_engine, _dev
            eng = get_engine(mol, None, dev)
            sh = eng.ao
            prim0 = np.cumsum([0] + [len(e) for e in sh.exps])[:-1].astype(np.int32)
            nprim = np.array([len(e) for e in sh.exps], np.int32)
            fn2sh = np.repeat(np.arange(sh.n, dtype=np.int32), 2 * sh.l + 1)
            self._cache[key] = dict(fn2sh=_dev(fn2sh, dev),
                eng=eng, nsh=sh.n, nao=sh.nao, l=_dev(sh.l, dev), ao0=_dev(sh.ao0, dev), prim0=_dev(prim0, dev),
                nprim=_dev(nprim, dev), exps=_dev(np.concatenate(sh.exps), dev),
                coefs=_dev(np.concatenate(sh.coefs), dev))
        return self._cache[key]

    def _grid_tables(self, grids, dev):
        import torch
        # keyed on the grid object's build counter (Grids.build / reset bump it), not only on id / size: a rebuild with
        # another partition scheme or geometry keeps the size but changes coords and weights
        key = ('grids', id(grids), getattr(grids, '_build_id', 0), grids.size, str(dev))
        if key not in self._cache:
            for k in [k for k in self._cache if k[0] == 'grids' and k[1] == id(grids)]:
                del self._cache[k]
            self._cache[key] = (torch.from_numpy(np.ascontiguousarray(grids.coords)).to(dev),
                                torch.from_numpy(np.ascontiguousarray(grids.weights)).to(dev))
        return self._cache[key]

    def reset(self):
        """Drop every device-side cache (shell tables, grid tables, sparse-grid plans)."""
        self._cache = {}
        return self

    def _call(self, name, fn, *args):
        if self.kernel_timer is not None:
            _lib_mod.check(self.kernel_timer.call(name, fn, *args))
        else:
            _lib_mod.check(fn(*args))
        if os.environ.get('PAMD_SYNC_DEBUG'):          # debugging aid: surface a device fault at the launch that caused it
            import torch
            torch.cuda.synchronize()
            print('[numint] %s ok' % name, flush=True)

    # -- value-based screening (the role of non0tab / pair_mask, numint.py:2845, eval_gto.py:146+) -------
    def _vmat_use_masks(self, mpanel, ng):
        """Screen the vmat GEMM only if the product of the two panel-tile densities leaves enough k-tiles out."""
        rows = (ng + 15) // 16
        if rows == 0:
            return False
        if self.vmat_screen_min_skip <= 0:
            return True                       # no density read-back (it would stall the launch queue)
        dens = float(mpanel[:rows].float().mean())
        return 1.0 - dens * dens >= self.vmat_screen_min_skip

    def _screen_masks(self, flags, ncomp, blk, ng):
        """AO tile flags [blk/16][ldao/16] -> (kmask for PAMD_orb_dot_rows, panel mask for PAMD_dgemm_tn_masked).
        One flag covers all components, so the scaled block aow = sum_c wv_c ao_c shares the panel mask."""
        nct = flags.shape[1]
        kmask = self._mask_n128_k16(flags, 1, blk)[:, :(ng + 127) // 128]
        kmask = kmask.expand(ncomp, kmask.shape[1], nct).contiguous()
        return kmask, self._mask_k16_n128(flags)

    @staticmethod
    def _mask_k16_n128(flags):
        """[rows/16][ld/16] -> [rows/16][ceil(ld/128)]: panel tiles of the grid-contracted GEMM."""
        import torch
        nrt, nct = flags.shape
        pad = (-nct) % 8
        if pad:
            flags = torch.nn.functional.pad(flags, (0, pad))
        return flags.view(nrt, -1, 8).amax(dim=2).contiguous()

    @staticmethod
    def _mask_n128_k16(flags, ncomp, blk):
        """[ncomp*blk/16][ld/16] -> [ncomp][blk/128][ld/16]: operand tiles of the AO-contracted GEMM."""
        nct = flags.shape[1]
        return flags.view(ncomp, blk // 128, 8, nct).amax(dim=2).contiguous()

    def eval_ao_block(self, mol, coords_dev, g0, ng, deriv, out, rows, ldao, flags=None):
        """out[comp][rows][ldao] <- AO values (deriv=0: comp=1; deriv=1: comp=4) for grid points [g0,g0+ng).
        flags (optional uint8 [rows/16][ldao/16], zeroed here): screening tile table filled by the same kernel."""
        import torch
        lib = _lib_mod.load_library()
        dev = coords_dev.device
        t = self._shell_tables(mol, dev)
        eng = t['eng']
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        if flags is not None:
            flags.zero_()
        self._call('eval_ao', lib.PAMD_eval_ao, _c.c_int(deriv), _ptr(t['l']), _ptr(t['ao0']), _ptr(t['prim0']),
                   _ptr(t['nprim']), _ptr(eng.ao_xyz), _ptr(t['exps']), _ptr(t['coefs']), _c.c_int(t['nsh']),
                   _ptr(t['fn2sh']), _c.c_int(t['nao']), _ptr(coords_dev), _c.c_long(g0), _c.c_long(ng), _ptr(eng.c2s),
                   _ptr(eng.c2s_off), _ptr(out), _c.c_long(rows), _c.c_int(ldao),
                   _c.c_double(self.screen_cutoff or 0.0), _ptr(flags) if flags is not None else _c.c_void_p(0), st)

    def eval_ao(self, mol, coords, deriv=0):
        """Host convenience (tests): (ngrids, nao), (4, ngrids, nao) or (10, ngrids, nao) like numint.eval_ao
        (pyscf/dft/numint.py:51-114)."""
        import torch
        dev = self._dev()
        c = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float64)).to(dev)
        ng = len(coords)
        nao = self._shell_tables(mol, dev)['nao']
        ldao = _round_up(nao, 16)
        ncomp = (1, 4, 10)[deriv]
        out = torch.zeros((ncomp, ng, ldao), dtype=torch.float64, device=dev)
        self.eval_ao_block(mol, c, 0, ng, deriv, out, ng, ldao)
        out = out[:, :, :nao].cpu().numpy()
        return out[0] if deriv == 0 else out

    # -- block-sparse pipeline ------------------------------------------------------------------------------------
    def sparse_plan(self, mol, grids, gga):
        """SparsePlan of (mol, grids, LDA|GGA) on this rank, cached on the content of mol and the build of grids."""
        from ..gto.moleintor import mol_fingerprint
        from .sparse_grid import SparsePlan
        dev = self._dev()
        rank, world = self._world()
        key = ('plan', mol_fingerprint(mol), id(grids), getattr(grids, '_build_id', 0), grids.size, int(bool(gga)),
               str(dev), rank, world, self.sparse_tile, self.sparse_cutoff)
        if key not in self._cache:
            for k in [k for k in self._cache if k[0] == 'plan']:      # one plan at a time: it may hold tens of GB
                del self._cache[k]
            plan = SparsePlan(self, mol, grids, gga, dev, rank, world)
            if plan.ao_c is not None:
                plan.release_dense()
            self._cache[key] = plan
        return self._cache[key]

    def _orbital_operand(self, dm, mo_coeff, mo_occ, nao, dev):
        """(orb_dev [rows][ldo], nocc, nocc_pad, ldo, sign_dev | None): occupied orbitals scaled by sqrt(occ) when the
        density is tagged (numint.py:2930-2994 _gen_rho_evaluator's MO branch), else the eigen-factorisation
        D_sym = C diag(sign) C^T of the symmetric part (a density only sees that part) - every density then goes
        through the same orbital-based kernels."""
        import torch
        sign = None
        if mo_coeff is not None:
            occ = np.asarray(mo_occ)
            orbo = np.asarray(mo_coeff)[:, occ > 0] * np.sqrt(occ[occ > 0])
        else:
            d = torch.from_numpy(np.ascontiguousarray((dm + dm.T) * .5)).to(dev)
            w, v = torch.linalg.eigh(d)
            thr = 1e-14 * max(float(w.abs().max()), 1e-300)
            keep = w.abs() > thr
            orbo = (v[:, keep] * w[keep].abs().sqrt()).cpu().numpy()
            sg = torch.sign(w[keep])
            if bool((sg < 0).any()):
                sign = sg.contiguous()
        nocc = orbo.shape[1]
        nocc_pad = _round_up(max(nocc, 1), 16)
        ldo = _round_up(nocc_pad, 160) if nocc_pad > 160 else nocc_pad
        orb_h = np.zeros((_round_up(nao + 1, 16), ldo))        # row nao (and beyond): zeros - padding columns gather it
        orb_h[:nao, :nocc] = orbo
        return torch.from_numpy(orb_h).to(dev), nocc, nocc_pad, ldo, sign

    def _sparse_xc(self, mol, grids, fac, gga, orbsets, spin, device_out=False):
        """nelec / exc / vmat of nr_rks (spin = 0, one orbital set) or nr_uks (spin = 1, two sets) on the compact AO
        subsets: per chunk of tiles  c = ao_c . C (sub_orb_dot) -> rho -> eval_xc -> aow_c (sub_scale) ->
        M[idx, idx] += ao_c^T aow_c (sub_vmat); finally V = M + M^T (numint.py:1157)."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        plan = self.sparse_plan(mol, grids, gga)
        G, ncomp, nao = plan.G, plan.ncomp, plan.nao
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        nset = len(orbsets)
        ldg = plan.max_chunk_points
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        M = torch.zeros((nset, nao, nao), dtype=f64, device=dev)
        acc = torch.zeros(3 if spin else 2, dtype=f64, device=dev)
        rho = torch.zeros((nset, 4, max(ldg, 1)), dtype=f64, device=dev)
        wv = torch.empty((nset, 4, max(ldg, 1)), dtype=f64, device=dev)
        nocc_pad_max = max(o[2] for o in orbsets)
        cmo = None                                     # allocated when the unfused orbital product runs (LDA, small nocc)
        aow = torch.empty(plan.max_aow_chunk + 256, dtype=f64, device=dev)       # sub_scale writes every element of a launch group;
        aow[plan.max_aow_chunk:].zero_()                                         # only the 256-double read slack needs defined values
        aoc_buf = None
        if plan.ao_c is None:
            aoc_buf = torch.zeros(plan.max_ao_chunk + 256, dtype=f64, device=dev)
        for ch in plan.chunks:
            t0, nt = ch['t0'], ch['t1'] - ch['t0']
            npts = nt * G
            if plan.ao_c is not None:
                aoc = plan.ao_c[ch['ao_base']:]
            else:
                plan.fill_chunk(ch, aoc_buf)
                aoc = aoc_buf
            tabs = (_ptr(plan.ao_off[t0:]), _ptr(plan.aow_off[t0:]), _ptr(plan.idx_off[t0:]), _ptr(plan.ld[t0:]))
            w_ch = plan.weights[t0 * G:(t0 + nt) * G]
            for s, (orb, nocc, nocc_pad, ldo, sign) in enumerate(orbsets):
                if nocc == 0:
                    rho[s].zero_()
                    continue
                if gga and self.fuse_rho:
                    # r04: rho / grad rho in the orbital product's epilogue - the c[comp][i][g] buffer (5.5 GB at config 3) is
                    # never written (PAMD_sub_orb_rho; returns 1 when the shape has no fused kernel)
                    if nocc_pad > 160:
                        rho[s].zero_()                     # several orbital chunks add up by atomics
                    args = (_ptr(aoc), tabs[0], tabs[2], tabs[3], _ptr(plan.idx), _c.c_int(nt), _c.c_int(G), _ptr(orb),
                            _c.c_int(ldo), _c.c_int(nocc), _c.c_int(nocc_pad), _ptr(sign) if sign is not None else _c.c_void_p(0),
                            _ptr(rho[s]), _c.c_long(ldg), st)
                    rc = (self.kernel_timer.call('ao_dot_mo', lib.PAMD_sub_orb_rho, *args) if self.kernel_timer is not None
                          else lib.PAMD_sub_orb_rho(*args))
                    if rc == 0:
                        continue
                    if rc < 0:
                        _lib_mod.check(rc)
                if cmo is None:
                    cmo = torch.empty(ncomp * nocc_pad_max * max(ldg, 1), dtype=f64, device=dev)
                self._call('ao_dot_mo', lib.PAMD_sub_orb_dot, _ptr(aoc), tabs[0], tabs[2], tabs[3], _ptr(plan.idx),
                           _c.c_int(nt), _c.c_int(G), _c.c_int(ncomp), _ptr(orb), _c.c_int(ldo), _c.c_int(nocc_pad),
                           _ptr(cmo), _c.c_long(nocc_pad * npts), _c.c_long(npts), st)
                self._call('rho', lib.PAMD_rho_from_mo, _ptr(cmo), _c.c_long(nocc_pad * npts), _c.c_long(npts),
                           _c.c_int(nocc), _c.c_int(ncomp), _c.c_long(npts), _ptr(rho[s]), _c.c_long(ldg),
                           _ptr(sign) if sign is not None else _c.c_void_p(0), st)
            if spin:
                self._call('eval_xc', lib.PAMD_eval_xc_pol, fac_c, _c.c_int(gga), _ptr(rho[0]), _ptr(rho[1]), _ptr(w_ch),
                           _c.c_long(npts), _c.c_long(ldg), _ptr(wv[0]), _ptr(wv[1]), _ptr(acc), _c.c_void_p(0), st)
            else:
                self._call('eval_xc', lib.PAMD_eval_xc, fac_c, _c.c_int(gga), _ptr(rho[0]), _ptr(w_ch), _c.c_long(npts),
                           _c.c_long(ldg), _ptr(wv[0]), _c.c_void_p(0), _ptr(acc), st)
            for s in range(nset):
                self._call('scale_ao', lib.PAMD_sub_scale_ao, _ptr(aoc), tabs[0], tabs[1], tabs[3], _c.c_int(nt),
                           _c.c_int(G), _c.c_int(ncomp), _c.c_int(ch['ld_max']), _ptr(wv[s]), _c.c_long(ldg), _ptr(aow), st)
                if self.vmat_sym:
                    self._call('ao_dot_aow', lib.PAMD_sub_vmat_sym, _ptr(aoc), tabs[0], _ptr(aow), tabs[1], tabs[2], tabs[3],
                               _ptr(plan.idx), _ptr(ch['work_sym']), _c.c_int(ch['nwork_sym']), _c.c_int(G), _c.c_int(nao),
                               _ptr(M[s]), _c.c_long(nao), st)
                else:
                    self._call('ao_dot_aow', lib.PAMD_sub_vmat, _ptr(aoc), tabs[0], _ptr(aow), tabs[1], tabs[2], tabs[3],
                               _ptr(plan.idx), _ptr(ch['work']), _c.c_int(ch['nwork']), _c.c_int(G), _c.c_int(nao),
                               _ptr(M[s]), _c.c_long(nao), st)
        v = torch.empty((nset, nao, nao), dtype=f64, device=dev)
        for s in range(nset):
            if self.vmat_sym:           # M holds the lower triangle of V = M + M^T (numint.py:1157) already
                self._call('reduce_sym', lib.PAMD_mirror_tril, _ptr(M[s]), _c.c_int(nao), _c.c_int(nao), _ptr(v[s]), st)
            else:
                self._call('reduce_sym', lib.PAMD_reduce_sym, _ptr(M[s]), _c.c_int(1), _c.c_int(nao), _c.c_int(nao),
                           _ptr(v[s]), st)
        rank, world = self._world()
        self._allreduce([v, acc], world)
        if device_out:
            return acc, v
        a_h, v_h = _lib_mod.download(self, [acc, v])
        return a_h, v_h

    def nr_rks_device(self, mol, grids, xc_code, orbo):
        """nr_rks for a closed-shell density given by its scaled occupied orbitals `orbo` = C_occ sqrt(occ), a (nao, nocc)
        DEVICE tensor; returns DEVICE tensors (acc = [nelec, exc], vmat (nao, nao)) - nothing crosses PCIe (the SCF loop of
        scf/device_scf.py).  Same arithmetic as nr_rks (numint.py:1074-1190) on the block-sparse pipeline."""
        import torch
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        nao, nocc = orbo.shape
        if xctype == 'HF':
            return torch.zeros(2, dtype=torch.float64, device=dev), torch.zeros((nao, nao), dtype=torch.float64, device=dev)
        gga = 1 if xctype == 'GGA' else 0
        nocc_pad = _round_up(max(nocc, 1), 16)
        ldo = _round_up(nocc_pad, 160) if nocc_pad > 160 else nocc_pad
        orb = torch.zeros((_round_up(nao + 1, 16), ldo), dtype=torch.float64, device=dev)
        orb[:nao, :nocc] = orbo
        acc, v = self._sparse_xc(mol, grids, fac, gga, [(orb, nocc, nocc_pad, ldo, None)], 0, device_out=True)
        return acc, v[0]

    def _first_order_terms(self, dms2, lowrank, nao, dev):
        """Per first-order density: ('pair', opA, opB, coef) when the caller tagged its factors (D = L R^T [+ h.c.], see
        df_jk._vk_lowrank), else ('op', operand) from the signed eigen-factorisation of the symmetric part."""
        import torch
        terms, cache = [], {}

        def operand(mat):
            key = id(mat)
            if key not in cache:
                m = np.ascontiguousarray(mat, dtype=np.float64)
                r = m.shape[1]
                rpad = _round_up(max(r, 1), 16)
                ldo = _round_up(rpad, 160) if rpad > 160 else rpad
                h = np.zeros((_round_up(nao + 1, 16), ldo))
                h[:nao, :r] = m
                cache[key] = (torch.from_numpy(h).to(dev), r, rpad, ldo, None)
            return cache[key]
        for k, d in enumerate(dms2):
            if lowrank is not None:
                lefts, rights, sym = lowrank
                terms.append(('pair', operand(lefts[k]), operand(rights[k]), 2.0 if sym else 1.0))
            else:
                terms.append(('op', self._orbital_operand(d, None, None, nao, dev)))
        return terms

    def _sparse_fxc(self, mol, grids, fac, gga, ops0, terms, spin):
        """vmat[s][i] of nr_rks_fxc (spin = 0: ops0 = [op], terms[i] = term) / nr_uks_fxc (spin = 1: ops0 = [op_a, op_b],
        terms[i] = (term_a, term_b)) on the compact AO subsets: rho0 and every rho1 from orbital products (sub_orb_dot),
        first-order weights from PAMD_eval_fxc / _pol, then the scale + scatter GEMM of the ground-state pipeline."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        plan = self.sparse_plan(mol, grids, gga)
        G, ncomp, nao = plan.G, plan.ncomp, plan.nao
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        nspin = 2 if spin else 1
        nvec = len(terms)
        ldg = max(plan.max_chunk_points, 1)
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        M = torch.zeros((nspin, nvec, nao, nao), dtype=f64, device=dev)
        rho0 = torch.zeros((nspin, 4, ldg), dtype=f64, device=dev)
        rho1 = torch.zeros((nspin, 4, ldg), dtype=f64, device=dev)
        wv = torch.empty((nspin, 4, ldg), dtype=f64, device=dev)
        all_ops = list(ops0)
        for t in terms:
            for tt in (t if spin else (t,)):
                all_ops += [tt[1]] if tt[0] == 'op' else [tt[1], tt[2]]
        pad_max = max(o[2] for o in all_ops)
        bufs = [torch.empty(ncomp * pad_max * ldg, dtype=f64, device=dev) for _ in range(3)]   # c0 / cached left / right
        aow = torch.empty(plan.max_aow_chunk + 256, dtype=f64, device=dev)       # sub_scale writes every element of a launch group;
        aow[plan.max_aow_chunk:].zero_()                                         # only the 256-double read slack needs defined values
        aoc_buf = torch.zeros(plan.max_ao_chunk + 256, dtype=f64, device=dev) if plan.ao_c is None else None
        for ch in plan.chunks:
            t0, nt = ch['t0'], ch['t1'] - ch['t0']
            npts = nt * G
            if plan.ao_c is not None:
                aoc = plan.ao_c[ch['ao_base']:]
            else:
                plan.fill_chunk(ch, aoc_buf)
                aoc = aoc_buf
            tabs = (_ptr(plan.ao_off[t0:]), _ptr(plan.aow_off[t0:]), _ptr(plan.idx_off[t0:]), _ptr(plan.ld[t0:]))
            w_ch = plan.weights[t0 * G:(t0 + nt) * G]

            def orb_dot(op, out):
                orb, nocc, nocc_pad, ldo, sign = op
                self._call('ao_dot_mo', lib.PAMD_sub_orb_dot, _ptr(aoc), tabs[0], tabs[2], tabs[3], _ptr(plan.idx),
                           _c.c_int(nt), _c.c_int(G), _c.c_int(ncomp), _ptr(orb), _c.c_int(ldo), _c.c_int(nocc_pad),
                           _ptr(out), _c.c_long(nocc_pad * npts), _c.c_long(npts), st)

            def density(op, out):
                orb, nocc, nocc_pad, ldo, sign = op
                if nocc == 0:
                    out.zero_()
                    return
                orb_dot(op, bufs[0])
                self._call('rho', lib.PAMD_rho_from_mo, _ptr(bufs[0]), _c.c_long(nocc_pad * npts), _c.c_long(npts),
                           _c.c_int(nocc), _c.c_int(ncomp), _c.c_long(npts), _ptr(out), _c.c_long(ldg),
                           _ptr(sign) if sign is not None else _c.c_void_p(0), st)
            for s in range(nspin):
                density(ops0[s], rho0[s])
            left_in_buf = None
            for i, term in enumerate(terms):
                for s, tt in enumerate(term if spin else (term,)):
                    if tt[0] == 'op':
                        density(tt[1], rho1[s])
                        continue
                    _, opa, opb, coef = tt
                    if opa[1] == 0:
                        rho1[s].zero_()
                        continue
                    if left_in_buf is not opa:           # densities of one reference share their left factor
                        orb_dot(opa, bufs[1])
                        left_in_buf = opa
                    orb_dot(opb, bufs[2])
                    self._call('rho', lib.PAMD_rho_from_mo_pair, _ptr(bufs[1]), _ptr(bufs[2]), _c.c_long(opa[2] * npts),
                               _c.c_long(npts), _c.c_int(opa[1]), _c.c_int(ncomp), _c.c_long(npts), _c.c_double(coef),
                               _ptr(rho1[s]), _c.c_long(ldg), st)
                if spin:
                    self._call('eval_fxc', lib.PAMD_eval_fxc_pol, fac_c, _c.c_int(gga), _ptr(rho0[0]), _ptr(rho0[1]),
                               _ptr(rho1[0]), _ptr(rho1[1]), _ptr(w_ch), _c.c_long(npts), _c.c_long(ldg), _ptr(wv[0]),
                               _ptr(wv[1]), st)
                else:
                    self._call('eval_fxc', lib.PAMD_eval_fxc, fac_c, _c.c_int(gga), _ptr(rho0[0]), _ptr(rho1[0]), _ptr(w_ch),
                               _c.c_long(npts), _c.c_long(ldg), _ptr(wv[0]), st)
                for s in range(nspin):
                    self._call('scale_ao', lib.PAMD_sub_scale_ao, _ptr(aoc), tabs[0], tabs[1], tabs[3], _c.c_int(nt),
                               _c.c_int(G), _c.c_int(ncomp), _c.c_int(ch['ld_max']), _ptr(wv[s]), _c.c_long(ldg), _ptr(aow), st)
                    if self.vmat_sym:
                        self._call('ao_dot_aow', lib.PAMD_sub_vmat_sym, _ptr(aoc), tabs[0], _ptr(aow), tabs[1], tabs[2], tabs[3],
                                   _ptr(plan.idx), _ptr(ch['work_sym']), _c.c_int(ch['nwork_sym']), _c.c_int(G), _c.c_int(nao),
                                   _ptr(M[s, i]), _c.c_long(nao), st)
                    else:
                        self._call('ao_dot_aow', lib.PAMD_sub_vmat, _ptr(aoc), tabs[0], _ptr(aow), tabs[1], tabs[2], tabs[3],
                                   _ptr(plan.idx), _ptr(ch['work']), _c.c_int(ch['nwork']), _c.c_int(G), _c.c_int(nao),
                                   _ptr(M[s, i]), _c.c_long(nao), st)
        v = torch.empty((nspin, nvec, nao, nao), dtype=f64, device=dev)
        for s in range(nspin):
            for i in range(nvec):
                if self.vmat_sym:
                    self._call('reduce_sym', lib.PAMD_mirror_tril, _ptr(M[s, i]), _c.c_int(nao), _c.c_int(nao), _ptr(v[s, i]), st)
                else:
                    self._call('reduce_sym', lib.PAMD_reduce_sym, _ptr(M[s, i]), _c.c_int(1), _c.c_int(nao), _c.c_int(nao),
                               _ptr(v[s, i]), st)
        rank, world = self._world()
        self._allreduce([v], world)
        return v.cpu().numpy()

    # -- the hot entry point ----------------------------------------------------------------------
    def nr_rks(self, mol, grids, xc_code, dms, relativity=0, hermi=1, max_memory=2000, verbose=None):
        """-> (nelec, excsum, vmat) with the contract of numint.nr_rks (numint.py:1074-1190)."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        dms_arr = np.asarray(dms)
        nao = dms_arr.shape[-1]
        shape = dms_arr.shape
        dms2 = dms_arr.reshape(-1, nao, nao)
        nset = len(dms2)
        mo_coeff = getattr(dms, 'mo_coeff', None)
        mo_occ = getattr(dms, 'mo_occ', None)
        nelec = np.zeros(nset)
        excsum = np.zeros(nset)
        if xctype == 'HF':
            vmat = np.zeros((nset, nao, nao))
            return (nelec[0], excsum[0], vmat[0]) if dms_arr.ndim == 2 else (nelec, excsum, vmat)
        gga = 1 if xctype == 'GGA' else 0
        ncomp = 4 if gga else 1
        if self.sparse:
            tagged = mo_coeff is not None and np.ndim(mo_occ) == 1 and nset == 1
            vs = []
            for iset in range(nset):
                ops = [self._orbital_operand(dms2[iset], mo_coeff if tagged else None, mo_occ, nao, dev)]
                a, v = self._sparse_xc(mol, grids, fac, gga, ops, 0)
                nelec[iset], excsum[iset] = a[0], a[1]
                vs.append(v[0])                 # the page-locked array lib.download handed out: no second 8 nao^2-byte host copy
            if dms_arr.ndim == 2:
                return nelec[0], excsum[0], vs[0]
            return nelec, excsum, (vs[0][None] if nset == 1 else np.stack(vs)).reshape(shape)
        vmat = np.zeros((nset, nao, nao))
        coords_dev, weights_dev = self._grid_tables(grids, dev)
        ngrids = grids.size
        ldao = _round_up(nao, 16)
        rank, world = self._world()
        blk = grid_block_size(ngrids, int(self.block_bytes // (ncomp * ldao * 8)), world)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        # +256 doubles of slack: the LDS-DMA GEMM reads whole 128-column panel rows
        ao = torch.zeros(ncomp * blk * ldao + 256, dtype=f64, device=dev)[:ncomp * blk * ldao].view(ncomp, blk, ldao)
        aow = torch.zeros(blk * ldao + 256, dtype=f64, device=dev)[:blk * ldao].view(blk, ldao)
        rho = torch.empty((4, blk), dtype=f64, device=dev)
        screen = self.screen_cutoff is not None
        fl_ao = torch.empty((blk // 16, ldao // 16), dtype=torch.uint8, device=dev) if screen else None
        wv = torch.empty((4, blk), dtype=f64, device=dev)
        nsplit = self.vmat_nsplit or pick_nsplit(((nao + 127) // 128) ** 2)
        nsplit_w = self.vmat_nsplit or pick_nsplit(((nao + 159) // 160) * ((nao + 127) // 128))
        nsplit_max = max(nsplit, nsplit_w)
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        for iset in range(nset):
            use_mo = mo_coeff is not None and np.ndim(mo_occ) == 1 and nset == 1
            if use_mo:
                occ = np.asarray(mo_occ)
                orbo = np.asarray(mo_coeff)[:, occ > 0] * np.sqrt(occ[occ > 0])
                nocc = orbo.shape[1]
                nocc_pad = _round_up(max(nocc, 1), 16)
                ldo = _round_up(nocc_pad, 160) if nocc_pad > 160 else nocc_pad
                orb_h = np.zeros((nao, ldo))
                orb_h[:, :nocc] = orbo
                orb = torch.from_numpy(orb_h).to(dev)
                cmo = torch.empty((ncomp, nocc_pad, blk), dtype=f64, device=dev)
            else:
                d = dms2[iset]
                ldd = _round_up(nao, 128)
                d_h = np.zeros((nao, ldd))
                d_h[:, :nao] = (d + d.T) * .5
                dsym = torch.from_numpy(d_h).to(dev)
                c0t = torch.empty((nao, blk), dtype=f64, device=dev)
            part = torch.zeros((nsplit_max, nao, nao), dtype=f64, device=dev)
            acc = torch.zeros(2, dtype=f64, device=dev)
            for ib, g0 in enumerate(range(0, ngrids, blk)):
                if ib % world != rank:
                    continue
                ng = min(blk, ngrids - g0)
                ng16 = _round_up(ng, 16)
                self.eval_ao_block(mol, coords_dev, g0, ng, gga, ao, blk, ldao, fl_ao)
                if screen:
                    kmask, mpanel = self._screen_masks(fl_ao, ncomp, blk, ng)
                if use_mo:
                    # c[comp][i][g] = sum_mu orb[mu][i] ao[comp][g][mu]   (GEMM shape of PAMD_cderi_solve)
                    self._call('ao_dot_mo', lib.PAMD_orb_dot_rows, _ptr(ao), _c.c_long(ldao), _c.c_long(blk * ldao),
                               _c.c_int(ncomp), _c.c_long(ng), _c.c_int(nao), _ptr(orb), _c.c_int(ldo),
                               _c.c_int(nocc_pad), _ptr(cmo), _c.c_long(blk),
                               _ptr(kmask) if screen else _c.c_void_p(0), st)
                    self._call('rho', lib.PAMD_rho_from_mo, _ptr(cmo), _c.c_long(nocc_pad * blk), _c.c_long(blk),
                               _c.c_int(nocc), _c.c_int(ncomp), _c.c_long(ng), _ptr(rho), _c.c_long(blk), _c.c_void_p(0), st)
                else:
                    # c0t[mu][g] = sum_nu D[nu][mu] ao0[g][nu]
                    self._call('dm_dot_ao', lib.PAMD_cderi_solve, _ptr(dsym), _c.c_int(ldd), _ptr(ao[0]),
                               _c.c_long(ldao), _ptr(c0t), _c.c_long(blk), _c.c_int(nao), _c.c_long(ng),
                               _c.c_int(nao), _c.c_int(0), _c.c_int(0), st)
                    self._call('rho', lib.PAMD_rho_from_dm, _ptr(ao), _ptr(c0t), _c.c_int(nao), _c.c_int(ldao),
                               _c.c_long(blk), _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _ptr(rho),
                               _c.c_long(blk), st)
                self._call('eval_xc', lib.PAMD_eval_xc, fac_c, _c.c_int(gga), _ptr(rho), _ptr(weights_dev[g0:g0 + ng]),
                           _c.c_long(ng), _c.c_long(blk), _ptr(wv), _c.c_void_p(0), _ptr(acc), st)
                self._call('scale_ao', lib.PAMD_scale_ao, _ptr(ao), _ptr(wv), _c.c_int(ldao), _c.c_long(blk),
                           _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _c.c_long(ng16), _ptr(aow), st)
                # vmat partial: M += ao0^T aow over the (16-padded, zero-weighted) grid rows of this block
                if screen and self._vmat_use_masks(mpanel, ng):
                    self._call('ao_dot_aow', lib.PAMD_dgemm_tn_masked, _ptr(ao[0]), _c.c_int(ldao), _ptr(aow),
                               _c.c_int(ldao), _ptr(part), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao),
                               _c.c_long(ng16), _c.c_int(nsplit), _ptr(mpanel), _ptr(mpanel), st)
                else:
                    self._call('ao_dot_aow', lib.PAMD_dgemm_tn, _ptr(ao[0]), _c.c_int(ldao), _ptr(aow),
                               _c.c_int(ldao), _ptr(part), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao),
                               _c.c_long(ng16), _c.c_int(2), _c.c_int(nsplit_w), st)
            v = torch.empty((nao, nao), dtype=f64, device=dev)
            self._call('reduce_sym', lib.PAMD_reduce_sym, _ptr(part), _c.c_int(nsplit_max), _c.c_int(nao), _c.c_int(nao),
                       _ptr(v), st)
            self._allreduce([v, acc], world)
            a = acc.cpu().numpy()
            nelec[iset], excsum[iset] = a[0], a[1]
            vmat[iset] = v.cpu().numpy()
        if dms_arr.ndim == 2:
            return nelec[0], excsum[0], vmat[0]
        return nelec, excsum, vmat.reshape(shape)

    def nr_vxc(self, mol, grids, xc_code, dms, spin=0, relativity=0, hermi=1, max_memory=2000, verbose=None):
        """numint.nr_vxc (numint.py:1052-1072): dispatch on spin to nr_rks / nr_uks."""
        fn = self.nr_rks if spin == 0 else self.nr_uks
        return fn(mol, grids, xc_code, dms, relativity, hermi, max_memory, verbose)

    def nr_rks_fxc(self, mol, grids, xc_code, dm0, dms, relativity=0, hermi=0, rho0=None, vxc=None, fxc=None,
                   max_memory=2000, verbose=None):
        """Closed-shell XC kernel contracted with first-order density matrices, the contract of numint.nr_rks_fxc
        (numint.py:1418-1530): vmat[i] = sum_g w [fxc : rho1_i] ao ao.  Per grid block: AO values once, rho0 from dm0 and
        rho1 from every dms[i] (``PAMD_rho_from_dm``; a density only sees the symmetric part of a matrix, so any hermi
        gives the same, symmetric, result), first-order weights from ``PAMD_eval_fxc`` (forward-over-forward AD of the
        same functional code as ``PAMD_eval_xc``), then the scale + GEMM of nr_rks.  The reference's cached rho0 / vxc /
        fxc arrays are accepted for signature compatibility and not used: the kernel is re-evaluated from dm0."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        dms_arr = np.asarray(dms)
        if np.iscomplexobj(dms_arr):
            raise NotImplementedError('complex density matrix')
        nao = dms_arr.shape[-1]
        shape = dms_arr.shape
        dms2 = dms_arr.reshape(-1, nao, nao)
        nset = len(dms2)
        vmat = np.zeros((nset, nao, nao))
        if xctype == 'HF':
            return vmat.reshape(shape)
        gga = 1 if xctype == 'GGA' else 0
        ncomp = 4 if gga else 1
        if self.sparse:
            dm0a = np.asarray(dm0, dtype=np.float64)
            tagged = getattr(dm0, 'mo_coeff', None) is not None and np.ndim(getattr(dm0, 'mo_occ', None)) == 1
            op0 = self._orbital_operand(dm0a, dm0.mo_coeff if tagged else None, dm0.mo_occ if tagged else None, nao, dev)
            terms = self._first_order_terms(dms2, getattr(dms, 'lowrank', None), nao, dev)
            return self._sparse_fxc(mol, grids, fac, gga, [op0], terms, 0)[0].reshape(shape)
        coords_dev, weights_dev = self._grid_tables(grids, dev)
        ngrids = grids.size
        ldao = _round_up(nao, 16)
        rank, world = self._world()
        blk = grid_block_size(ngrids, int(self.block_bytes // (ncomp * ldao * 8)), world)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        ao = torch.zeros(ncomp * blk * ldao + 256, dtype=f64, device=dev)[:ncomp * blk * ldao].view(ncomp, blk, ldao)
        aow = torch.zeros(blk * ldao + 256, dtype=f64, device=dev)[:blk * ldao].view(blk, ldao)
        rho_0 = torch.zeros((4, blk), dtype=f64, device=dev)
        rho_1 = torch.zeros((4, blk), dtype=f64, device=dev)
        wv = torch.empty((4, blk), dtype=f64, device=dev)
        c0t = torch.empty((nao, blk), dtype=f64, device=dev)
        nsplit = self.vmat_nsplit or pick_nsplit(((nao + 159) // 160) * ((nao + 127) // 128))
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        ldd = _round_up(nao, 128)

        def padded(d):
            d_h = np.zeros((nao, ldd))
            d_h[:, :nao] = (d + d.T) * .5
            return torch.from_numpy(d_h).to(dev)
        d0 = padded(np.asarray(dm0, dtype=np.float64))
        d1 = [padded(d) for d in dms2]
        parts = torch.zeros((nset, nsplit, nao, nao), dtype=f64, device=dev)

        def density(dmat, ng, out):
            self._call('dm_dot_ao', lib.PAMD_cderi_solve, _ptr(dmat), _c.c_int(ldd), _ptr(ao[0]), _c.c_long(ldao),
                       _ptr(c0t), _c.c_long(blk), _c.c_int(nao), _c.c_long(ng), _c.c_int(nao), _c.c_int(0), _c.c_int(0), st)
            self._call('rho', lib.PAMD_rho_from_dm, _ptr(ao), _ptr(c0t), _c.c_int(nao), _c.c_int(ldao), _c.c_long(blk),
                       _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _ptr(out), _c.c_long(blk), st)
        for ib, g0 in enumerate(range(0, ngrids, blk)):
            if ib % world != rank:
                continue
            ng = min(blk, ngrids - g0)
            ng16 = _round_up(ng, 16)
            self.eval_ao_block(mol, coords_dev, g0, ng, gga, ao, blk, ldao, None)
            density(d0, ng, rho_0)
            for i in range(nset):
                density(d1[i], ng, rho_1)
                self._call('eval_fxc', lib.PAMD_eval_fxc, fac_c, _c.c_int(gga), _ptr(rho_0), _ptr(rho_1),
                           _ptr(weights_dev[g0:g0 + ng]), _c.c_long(ng), _c.c_long(blk), _ptr(wv), st)
                self._call('scale_ao', lib.PAMD_scale_ao, _ptr(ao), _ptr(wv), _c.c_int(ldao), _c.c_long(blk),
                           _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _c.c_long(ng16), _ptr(aow), st)
                self._call('ao_dot_aow', lib.PAMD_dgemm_tn, _ptr(ao[0]), _c.c_int(ldao), _ptr(aow), _c.c_int(ldao),
                           _ptr(parts[i]), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao), _c.c_long(ng16), _c.c_int(2),
                           _c.c_int(nsplit), st)
        v = torch.empty((nao, nao), dtype=f64, device=dev)
        for i in range(nset):
            self._call('reduce_sym', lib.PAMD_reduce_sym, _ptr(parts[i]), _c.c_int(nsplit), _c.c_int(nao), _c.c_int(nao),
                       _ptr(v), st)
            self._allreduce([v], world)
            vmat[i] = v.cpu().numpy()
        return vmat.reshape(shape)

    def nr_uks_fxc(self, mol, grids, xc_code, dm0, dms, relativity=0, hermi=0, rho0=None, vxc=None, fxc=None,
                   max_memory=2000, verbose=None):
        """Spin-polarised XC kernel contracted with first-order spin density matrices, the contract of numint.nr_uks_fxc
        (numint.py:1690-1832): dm0 = (dm0_alpha, dm0_beta); dms = (dm1_alpha, dm1_beta), each (nao, nao) or
        (nset, nao, nao); returns vmat (2, nao, nao) or (2, nset, nao, nao).  Same pipeline as nr_rks_fxc with
        ``PAMD_eval_fxc_pol``; cached rho0 / vxc / fxc are not used."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        dma, dmb = np.asarray(dms[0], dtype=np.float64), np.asarray(dms[1], dtype=np.float64)
        nao = dma.shape[-1]
        single = dma.ndim == 2
        dma, dmb = dma.reshape(-1, nao, nao), dmb.reshape(-1, nao, nao)
        nset = len(dma)
        vmat = np.zeros((2, nset, nao, nao))
        if xctype != 'HF' and self.sparse:
            gga = 1 if xctype == 'GGA' else 0
            mo0, occ0 = getattr(dm0, 'mo_coeff', None), getattr(dm0, 'mo_occ', None)
            tagged = mo0 is not None and np.ndim(occ0) == 2
            ops0 = [self._orbital_operand(np.asarray(dm0[s], dtype=np.float64), np.asarray(mo0[s]) if tagged else None,
                                          np.asarray(occ0[s]) if tagged else None, nao, dev) for s in range(2)]
            lr = getattr(dms, 'lowrank', None)          # factors listed alpha densities first, then beta
            if lr is not None:
                lr_a = (lr[0][:nset], lr[1][:nset], lr[2])
                lr_b = (lr[0][nset:], lr[1][nset:], lr[2])
            else:
                lr_a = lr_b = None
            ta = self._first_order_terms(dma, lr_a, nao, dev)
            tb = self._first_order_terms(dmb, lr_b, nao, dev)
            vmat = self._sparse_fxc(mol, grids, fac, gga, ops0, list(zip(ta, tb)), 1)
            return vmat[:, 0] if single else vmat
        if xctype != 'HF':
            gga = 1 if xctype == 'GGA' else 0
            ncomp = 4 if gga else 1
            coords_dev, weights_dev = self._grid_tables(grids, dev)
            ngrids = grids.size
            ldao = _round_up(nao, 16)
            rank, world = self._world()
            blk = grid_block_size(ngrids, int(self.block_bytes // (ncomp * ldao * 8)), world)
            st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
            f64 = torch.float64
            ao = torch.zeros(ncomp * blk * ldao + 256, dtype=f64, device=dev)[:ncomp * blk * ldao].view(ncomp, blk, ldao)
            aow = torch.zeros(blk * ldao + 256, dtype=f64, device=dev)[:blk * ldao].view(blk, ldao)
            rho_0 = torch.zeros((2, 4, blk), dtype=f64, device=dev)
            rho_1 = torch.zeros((2, 4, blk), dtype=f64, device=dev)
            wv = torch.empty((2, 4, blk), dtype=f64, device=dev)
            c0t = torch.empty((nao, blk), dtype=f64, device=dev)
            nsplit = self.vmat_nsplit or pick_nsplit(((nao + 159) // 160) * ((nao + 127) // 128))
            fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
            ldd = _round_up(nao, 128)

            def padded(d):
                d_h = np.zeros((nao, ldd))
                d_h[:, :nao] = (d + d.T) * .5
                return torch.from_numpy(d_h).to(dev)
            d0 = [padded(np.asarray(dm0[s], dtype=np.float64)) for s in range(2)]
            d1 = [[padded(d) for d in dma], [padded(d) for d in dmb]]
            parts = torch.zeros((2, nset, nsplit, nao, nao), dtype=f64, device=dev)

            def density(dmat, ng, out):
                self._call('dm_dot_ao', lib.PAMD_cderi_solve, _ptr(dmat), _c.c_int(ldd), _ptr(ao[0]), _c.c_long(ldao),
                           _ptr(c0t), _c.c_long(blk), _c.c_int(nao), _c.c_long(ng), _c.c_int(nao), _c.c_int(0), _c.c_int(0), st)
                self._call('rho', lib.PAMD_rho_from_dm, _ptr(ao), _ptr(c0t), _c.c_int(nao), _c.c_int(ldao), _c.c_long(blk),
                           _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _ptr(out), _c.c_long(blk), st)
            for ib, g0 in enumerate(range(0, ngrids, blk)):
                if ib % world != rank:
                    continue
                ng = min(blk, ngrids - g0)
                ng16 = _round_up(ng, 16)
                self.eval_ao_block(mol, coords_dev, g0, ng, gga, ao, blk, ldao, None)
                for s in range(2):
                    density(d0[s], ng, rho_0[s])
                for i in range(nset):
                    for s in range(2):
                        density(d1[s][i], ng, rho_1[s])
                    self._call('eval_fxc', lib.PAMD_eval_fxc_pol, fac_c, _c.c_int(gga), _ptr(rho_0[0]), _ptr(rho_0[1]),
                               _ptr(rho_1[0]), _ptr(rho_1[1]), _ptr(weights_dev[g0:g0 + ng]), _c.c_long(ng), _c.c_long(blk),
                               _ptr(wv[0]), _ptr(wv[1]), st)
                    for s in range(2):
                        self._call('scale_ao', lib.PAMD_scale_ao, _ptr(ao), _ptr(wv[s]), _c.c_int(ldao), _c.c_long(blk),
                                   _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _c.c_long(ng16), _ptr(aow), st)
                        self._call('ao_dot_aow', lib.PAMD_dgemm_tn, _ptr(ao[0]), _c.c_int(ldao), _ptr(aow), _c.c_int(ldao),
                                   _ptr(parts[s, i]), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao), _c.c_long(ng16),
                                   _c.c_int(2), _c.c_int(nsplit), st)
            v = torch.empty((nao, nao), dtype=f64, device=dev)
            for s in range(2):
                for i in range(nset):
                    self._call('reduce_sym', lib.PAMD_reduce_sym, _ptr(parts[s, i]), _c.c_int(nsplit), _c.c_int(nao),
                               _c.c_int(nao), _ptr(v), st)
                    self._allreduce([v], world)
                    vmat[s, i] = v.cpu().numpy()
        return vmat[:, 0] if single else vmat

    def nr_rks_fxc_st(self, mol, grids, xc_code, dm0, dms_alpha, relativity=0, hermi=0, singlet=True, rho0=None, vxc=None,
                      fxc=None, max_memory=2000, verbose=None):
        """Singlet / triplet kernel of a closed-shell reference contracted with ALPHA first-order density matrices
        (numint.py:1532-1549: fxc_aa +- fxc_ab): the alpha response of the spin-polarised kernel at (dm0/2, dm0/2) to
        (dm1, +dm1) or (dm1, -dm1)."""
        half = np.asarray(dm0, dtype=np.float64) * .5
        d1 = np.asarray(dms_alpha, dtype=np.float64)
        dm0_ab, dms_ab = (half, half), (d1, d1 if singlet else -d1)
        mo0, occ0 = getattr(dm0, 'mo_coeff', None), getattr(dm0, 'mo_occ', None)
        if mo0 is not None and np.ndim(occ0) == 1:          # keep the orbital tag: (c, c) with half occupations
            dm0_ab = _lib_mod.tag_array(np.array(dm0_ab), mo_coeff=np.array([mo0, mo0]),
                                        mo_occ=np.array([occ0, occ0]) * .5)
        lr = getattr(dms_alpha, 'lowrank', None)
        if lr is not None:                                  # alpha factors, then beta factors (+- the same)
            lefts, rights, sym = lr
            sgn = 1.0 if singlet else -1.0
            dms_ab = _lib_mod.tag_array(np.array(dms_ab), lowrank=(list(lefts) * 2,
                                                                   list(rights) + [r * sgn for r in rights], sym))
        return self.nr_uks_fxc(mol, grids, xc_code, dm0_ab, dms_ab, relativity, hermi, max_memory=max_memory)[0]

    def nr_fxc(self, mol, grids, xc_code, dm0, dms, spin=0, relativity=0, hermi=0, rho0=None, vxc=None, fxc=None,
               max_memory=2000, verbose=None):
        """numint.nr_fxc (numint.py:2846-2860): dispatch on spin."""
        fn = self.nr_rks_fxc if spin == 0 else self.nr_uks_fxc
        return fn(mol, grids, xc_code, dm0, dms, relativity, hermi, rho0, vxc, fxc, max_memory, verbose)

    def nr_rks_grad(self, mol, grids, xc_code, dm, grid_response=False):
        """XC part of the closed-shell nuclear gradient, (natm, 3), grid response left out: the contraction
        -2 sum_{mu on A, nu} vmat[x]_{mu nu} D_{mu nu} of pyscf/grad/rks.py:get_vxc (:197-255; _d1_dot_,
        _gga_grad_sum_, _make_dR_dao_w) done per grid block on the device without forming vmat[x]."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        natm = mol.natm
        if xctype == 'HF':
            return np.zeros((natm, 3))
        gga = 1 if xctype == 'GGA' else 0
        nao = mol.nao_nr()
        ldao = _round_up(nao, 16)
        ncomp_ao, ncomp_c = (10, 4) if gga else (4, 1)
        coords_dev, weights_dev = self._grid_tables(grids, dev)
        ngrids = grids.size
        rank, world = self._world()
        blk = grid_block_size(ngrids, int(self.block_bytes // ((ncomp_ao + ncomp_c) * ldao * 8)), world)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        ao = torch.zeros((ncomp_ao, blk, ldao), dtype=f64, device=dev)
        c = torch.zeros((ncomp_c, blk, ldao), dtype=f64, device=dev)
        rho = torch.zeros((4, blk), dtype=f64, device=dev)
        wv = torch.empty((4, blk), dtype=f64, device=dev)
        acc = torch.zeros(2, dtype=f64, device=dev)
        out = torch.zeros((3, nao), dtype=f64, device=dev)
        resp = self._response_setup(mol, grids, dev, blk) if grid_response else None
        d_h = np.zeros((nao, ldao))
        d_h[:, :nao] = (np.asarray(dm) + np.asarray(dm).T) * .5
        dsym = torch.from_numpy(d_h).to(dev)
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        for ib, g0 in enumerate(range(0, ngrids, blk)):
            if ib % world != rank:
                continue
            ng = min(blk, ngrids - g0)
            self.eval_ao_block(mol, coords_dev, g0, ng, 2 if gga else 1, ao, blk, ldao)
            c.zero_()
            for k in range(ncomp_c):        # c_k[g][mu] = sum_nu ao_k[g][nu] D[mu][nu]
                self._call('ao_dot_dm', lib.PAMD_dgemm_nt, _ptr(ao[k]), _c.c_long(ldao), _ptr(dsym), _c.c_long(ldao),
                           _ptr(c[k]), _c.c_int(ldao), _c.c_int(ng), _c.c_int(nao), _c.c_long(nao), _c.c_int(1), st)
            rho[0, :ng] = (ao[0, :ng] * c[0, :ng]).sum(dim=1)
            if gga:
                for k in range(1, 4):
                    rho[k, :ng] = 2 * (ao[k, :ng] * c[0, :ng]).sum(dim=1)
            self._call('eval_xc', lib.PAMD_eval_xc, fac_c, _c.c_int(gga), _ptr(rho), _ptr(weights_dev[g0:g0 + ng]),
                       _c.c_long(ng), _c.c_long(blk), _ptr(wv), _ptr(resp['exc']) if resp else _c.c_void_p(0),
                       _ptr(acc), st)
            self._call('xc_grad', lib.PAMD_xc_grad, _ptr(ao), _ptr(c), _ptr(wv), _c.c_int(ldao), _c.c_long(blk),
                       _c.c_long(blk), _c.c_int(gga), _c.c_long(ng), _c.c_int(nao), _ptr(out), st)
            if resp:
                resp['evol'][:ng] = resp['exc'][:ng] * rho[0, :ng]             # per particle -> per volume
                self._response_block(resp, coords_dev, weights_dev, g0, ng, [(ao, c, wv)], gga, ldao, blk, nao, st)
        self._allreduce([out], world)
        s = out.cpu().numpy()
        aoslices = mol.aoslice_by_atom()
        de = np.zeros((natm, 3))
        for ia in range(natm):
            p0, p1 = aoslices[ia][2], aoslices[ia][3]
            de[ia] = -2 * s[:, p0:p1].sum(axis=1)
        if resp:
            de += self._response_finish(resp, world)
        return de

    # -- grid response of the XC gradient (pyscf/grad/rks.py:257-340) -------------------------------------------
    def _response_setup(self, mol, grids, dev, blk):
        import torch
        natm = mol.natm
        from . import gen_grid
        scheme = gen_grid.scheme_id(grids.becke_scheme)
        owner = np.zeros(grids.size, np.int32)
        owner[:len(grids.atm_idx)] = grids.atm_idx                    # alignment padding: weight 0, any owner
        table = None
        if callable(grids.radii_adjust) and grids.atomic_radii is not None:
            table = grids.radii_adjust(mol, grids.atomic_radii)
        f64 = torch.float64
        return dict(natm=natm, scheme=scheme, owner=torch.from_numpy(owner).to(dev),
                    atm=torch.from_numpy(np.ascontiguousarray(mol.atom_coords())).to(dev),
                    table=None if table is None else torch.from_numpy(np.ascontiguousarray(table)).to(dev),
                    pb=torch.empty((natm, blk), dtype=f64, device=dev), exc=torch.zeros(blk, dtype=f64, device=dev),
                    evol=torch.zeros(blk, dtype=f64, device=dev), rows=torch.zeros((3, blk), dtype=f64, device=dev),
                    de_w=torch.zeros((natm, 3), dtype=f64, device=dev), de_move=torch.zeros((natm, 3), dtype=f64, device=dev))

    def _response_block(self, resp, coords_dev, weights_dev, g0, ng, operands, gga, ldao, blk, nao, st):
        """Weight-derivative term (PAMD_becke_response) and the points' own motion (PAMD_xc_grad_rows, summed per owner
        atom) for grid points [g0, g0 + ng); operands = [(ao, c, wv)] per spin."""
        lib = _lib_mod.load_library()
        natm = resp['natm']
        pbv = resp['pb'].view(-1)[:natm * ng].view(natm, ng)
        tptr = _ptr(resp['table']) if resp['table'] is not None else _c.c_void_p(0)
        self._call('becke', lib.PAMD_grid_partition, _ptr(pbv), _ptr(coords_dev[g0:g0 + ng]), _ptr(resp['atm']), tptr,
                   _c.c_int(natm), _c.c_long(ng), _c.c_int(resp['scheme']), st)
        self._call('becke_response', lib.PAMD_grid_response, _ptr(coords_dev[g0:g0 + ng]), _ptr(resp['owner'][g0:g0 + ng]),
                   _ptr(weights_dev[g0:g0 + ng]), _ptr(resp['evol']), _ptr(pbv), _ptr(resp['atm']), tptr,
                   _c.c_int(natm), _c.c_long(ng), _c.c_int(resp['scheme']), _ptr(resp['de_w']), st)
        resp['rows'].zero_()
        for ao, c, wv in operands:
            self._call('xc_grad_rows', lib.PAMD_xc_grad_rows, _ptr(ao), _ptr(c), _ptr(wv), _c.c_int(ldao), _c.c_long(blk),
                       _c.c_long(blk), _c.c_int(gga), _c.c_long(ng), _c.c_int(nao), _ptr(resp['rows']), st)
        # per-owner sums as a small dense product (torch's index_add_ faults on this ROCm build for ng > ~30 k)
        import torch
        own = resp['owner'][g0:g0 + ng].long()
        onehot = (own[:, None] == torch.arange(natm, device=own.device)[None, :]).to(torch.float64)
        resp['de_move'] += onehot.T @ resp['rows'][:, :ng].T

    def _response_finish(self, resp, world):
        tot = resp['de_w'] + 2 * resp['de_move']
        self._allreduce([tot], world)
        return tot.cpu().numpy()

    def nr_uks_grad(self, mol, grids, xc_code, dms, grid_response=False):
        """Spin-polarised XC nuclear gradient (natm, 3), grid response left out (pyscf/grad/uks.py:get_vxc
        :100-190 contracted with (D_alpha, D_beta)); same device pipeline as nr_rks_grad, two spin passes per block."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        natm = mol.natm
        if xctype == 'HF':
            return np.zeros((natm, 3))
        gga = 1 if xctype == 'GGA' else 0
        nao = mol.nao_nr()
        ldao = _round_up(nao, 16)
        ncomp_ao, ncomp_c = (10, 4) if gga else (4, 1)
        coords_dev, weights_dev = self._grid_tables(grids, dev)
        ngrids = grids.size
        rank, world = self._world()
        blk = grid_block_size(ngrids, int(self.block_bytes // ((ncomp_ao + 2 * ncomp_c) * ldao * 8)), world)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        ao = torch.zeros((ncomp_ao, blk, ldao), dtype=f64, device=dev)
        c = torch.zeros((2, ncomp_c, blk, ldao), dtype=f64, device=dev)
        rho = torch.zeros((2, 4, blk), dtype=f64, device=dev)
        wv = torch.empty((2, 4, blk), dtype=f64, device=dev)
        acc = torch.zeros(3, dtype=f64, device=dev)
        out = torch.zeros((3, nao), dtype=f64, device=dev)
        resp = self._response_setup(mol, grids, dev, blk) if grid_response else None
        evol = resp['evol'] if resp else None
        dsym = []
        for s in range(2):
            d_h = np.zeros((nao, ldao))
            d_h[:, :nao] = (np.asarray(dms[s]) + np.asarray(dms[s]).T) * .5
            dsym.append(torch.from_numpy(d_h).to(dev))
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        for ib, g0 in enumerate(range(0, ngrids, blk)):
            if ib % world != rank:
                continue
            ng = min(blk, ngrids - g0)
            self.eval_ao_block(mol, coords_dev, g0, ng, 2 if gga else 1, ao, blk, ldao)
            c.zero_()
            for s in range(2):
                for k in range(ncomp_c):
                    self._call('ao_dot_dm', lib.PAMD_dgemm_nt, _ptr(ao[k]), _c.c_long(ldao), _ptr(dsym[s]), _c.c_long(ldao),
                               _ptr(c[s, k]), _c.c_int(ldao), _c.c_int(ng), _c.c_int(nao), _c.c_long(nao), _c.c_int(1), st)
                rho[s, 0, :ng] = (ao[0, :ng] * c[s, 0, :ng]).sum(dim=1)
                if gga:
                    for k in range(1, 4):
                        rho[s, k, :ng] = 2 * (ao[k, :ng] * c[s, 0, :ng]).sum(dim=1)
            self._call('eval_xc', lib.PAMD_eval_xc_pol, fac_c, _c.c_int(gga), _ptr(rho[0]), _ptr(rho[1]),
                       _ptr(weights_dev[g0:g0 + ng]), _c.c_long(ng), _c.c_long(blk), _ptr(wv[0]), _ptr(wv[1]),
                       _ptr(acc), _ptr(evol) if evol is not None else _c.c_void_p(0), st)
            for s in range(2):
                self._call('xc_grad', lib.PAMD_xc_grad, _ptr(ao), _ptr(c[s]), _ptr(wv[s]), _c.c_int(ldao), _c.c_long(blk),
                           _c.c_long(blk), _c.c_int(gga), _c.c_long(ng), _c.c_int(nao), _ptr(out), st)
            if resp:
                self._response_block(resp, coords_dev, weights_dev, g0, ng, [(ao, c[0], wv[0]), (ao, c[1], wv[1])], gga,
                                     ldao, blk, nao, st)
        self._allreduce([out], world)
        sv = out.cpu().numpy()
        aoslices = mol.aoslice_by_atom()
        de = np.zeros((natm, 3))
        for ia in range(natm):
            p0, p1 = aoslices[ia][2], aoslices[ia][3]
            de[ia] = -2 * sv[:, p0:p1].sum(axis=1)
        if resp:
            de += self._response_finish(resp, world)
        return de

    def nr_uks(self, mol, grids, xc_code, dms, relativity=0, hermi=1, max_memory=2000, verbose=None):
        """-> (nelec[2], excsum, vmat[2]) with the contract of numint.nr_uks (numint.py:1192-1324);
        dms = (dm_alpha, dm_beta), optionally tagged with mo_coeff (2, nao, nmo) / mo_occ (2, nmo)."""
        import torch
        lib = _lib_mod.load_library()
        dev = self._dev()
        if grids.coords is None:
            grids.build()
        hyb, fac = self._parse(xc_code)
        xctype = _xc.xc_type(xc_code)
        dms_arr = np.asarray(dms)
        assert dms_arr.ndim == 3 and dms_arr.shape[0] == 2, 'nr_uks expects (dm_alpha, dm_beta)'
        nao = dms_arr.shape[-1]
        if xctype == 'HF':
            return np.zeros(2), 0.0, np.zeros((2, nao, nao))
        gga = 1 if xctype == 'GGA' else 0
        ncomp = 4 if gga else 1
        mo_coeff = getattr(dms, 'mo_coeff', None)
        mo_occ = getattr(dms, 'mo_occ', None)
        use_mo = mo_coeff is not None and np.ndim(mo_occ) == 2
        if self.sparse:
            ops = [self._orbital_operand(dms_arr[s], np.asarray(mo_coeff[s]) if use_mo else None,
                                         np.asarray(mo_occ[s]) if use_mo else None, nao, dev) for s in range(2)]
            a, v = self._sparse_xc(mol, grids, fac, gga, ops, 1)
            return a[:2].copy(), float(a[2]), v
        coords_dev, weights_dev = self._grid_tables(grids, dev)
        ngrids = grids.size
        ldao = _round_up(nao, 16)
        rank, world = self._world()
        blk = grid_block_size(ngrids, int(self.block_bytes // (ncomp * ldao * 8)), world)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        f64 = torch.float64
        ao = torch.zeros(ncomp * blk * ldao + 256, dtype=f64, device=dev)[:ncomp * blk * ldao].view(ncomp, blk, ldao)
        aow = torch.zeros(blk * ldao + 256, dtype=f64, device=dev)[:blk * ldao].view(blk, ldao)
        rho = torch.zeros((2, 4, blk), dtype=f64, device=dev)
        screen = self.screen_cutoff is not None
        fl_ao = torch.empty((blk // 16, ldao // 16), dtype=torch.uint8, device=dev) if screen else None
        wv = torch.empty((2, 4, blk), dtype=f64, device=dev)
        nsplit = self.vmat_nsplit or pick_nsplit(((nao + 127) // 128) ** 2)
        nsplit_w = self.vmat_nsplit or pick_nsplit(((nao + 159) // 160) * ((nao + 127) // 128))
        nsplit_max = max(nsplit, nsplit_w)
        fac_c = (ctypes.c_double * _xc.NFAC)(*fac)
        ops = []
        for s in range(2):
            if use_mo:
                occ = np.asarray(mo_occ[s])
                orbo = np.asarray(mo_coeff[s])[:, occ > 0] * np.sqrt(occ[occ > 0])
                nocc = orbo.shape[1]
                nocc_pad = _round_up(max(nocc, 1), 16)
                ldo = _round_up(nocc_pad, 160) if nocc_pad > 160 else nocc_pad
                orb_h = np.zeros((nao, ldo))
                orb_h[:, :nocc] = orbo
                ops.append((torch.from_numpy(orb_h).to(dev), nocc, nocc_pad, ldo,
                            torch.empty((ncomp, nocc_pad, blk), dtype=f64, device=dev)))
            else:
                d = dms_arr[s]
                ldd = _round_up(nao, 128)
                d_h = np.zeros((nao, ldd))
                d_h[:, :nao] = (d + d.T) * .5
                ops.append((torch.from_numpy(d_h).to(dev), ldd, torch.empty((nao, blk), dtype=f64, device=dev)))
        part = torch.zeros((2, nsplit_max, nao, nao), dtype=f64, device=dev)
        acc = torch.zeros(3, dtype=f64, device=dev)
        evol = None
        for ib, g0 in enumerate(range(0, ngrids, blk)):
            if ib % world != rank:
                continue
            ng = min(blk, ngrids - g0)
            ng16 = _round_up(ng, 16)
            self.eval_ao_block(mol, coords_dev, g0, ng, gga, ao, blk, ldao, fl_ao)
            if screen:
                kmask, mpanel = self._screen_masks(fl_ao, ncomp, blk, ng)
            for s in range(2):
                if use_mo:
                    orb, nocc, nocc_pad, ldo, cmo = ops[s]
                    if nocc == 0:
                        rho[s].zero_()
                        continue
                    self._call('ao_dot_mo', lib.PAMD_orb_dot_rows, _ptr(ao), _c.c_long(ldao), _c.c_long(blk * ldao),
                               _c.c_int(ncomp), _c.c_long(ng), _c.c_int(nao), _ptr(orb), _c.c_int(ldo),
                               _c.c_int(nocc_pad), _ptr(cmo), _c.c_long(blk),
                               _ptr(kmask) if screen else _c.c_void_p(0), st)
                    self._call('rho', lib.PAMD_rho_from_mo, _ptr(cmo), _c.c_long(nocc_pad * blk), _c.c_long(blk),
                               _c.c_int(nocc), _c.c_int(ncomp), _c.c_long(ng), _ptr(rho[s]), _c.c_long(blk), _c.c_void_p(0),
                               st)
                else:
                    dsym, ldd, c0t = ops[s]
                    self._call('dm_dot_ao', lib.PAMD_cderi_solve, _ptr(dsym), _c.c_int(ldd), _ptr(ao[0]),
                               _c.c_long(ldao), _ptr(c0t), _c.c_long(blk), _c.c_int(nao), _c.c_long(ng),
                               _c.c_int(nao), _c.c_int(0), _c.c_int(0), st)
                    self._call('rho', lib.PAMD_rho_from_dm, _ptr(ao), _ptr(c0t), _c.c_int(nao), _c.c_int(ldao),
                               _c.c_long(blk), _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _ptr(rho[s]),
                               _c.c_long(blk), st)
            self._call('eval_xc', lib.PAMD_eval_xc_pol, fac_c, _c.c_int(gga), _ptr(rho[0]), _ptr(rho[1]),
                       _ptr(weights_dev[g0:g0 + ng]), _c.c_long(ng), _c.c_long(blk), _ptr(wv[0]), _ptr(wv[1]),
                       _ptr(acc), _ptr(evol) if evol is not None else _c.c_void_p(0), st)
            use_masks = screen and self._vmat_use_masks(mpanel, ng)
            for s in range(2):
                self._call('scale_ao', lib.PAMD_scale_ao, _ptr(ao), _ptr(wv[s]), _c.c_int(ldao), _c.c_long(blk),
                           _c.c_long(blk), _c.c_int(ncomp), _c.c_long(ng), _c.c_long(ng16), _ptr(aow), st)
                if screen and use_masks:
                    self._call('ao_dot_aow', lib.PAMD_dgemm_tn_masked, _ptr(ao[0]), _c.c_int(ldao), _ptr(aow),
                               _c.c_int(ldao), _ptr(part[s]), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao),
                               _c.c_long(ng16), _c.c_int(nsplit), _ptr(mpanel), _ptr(mpanel), st)
                else:
                    self._call('ao_dot_aow', lib.PAMD_dgemm_tn, _ptr(ao[0]), _c.c_int(ldao), _ptr(aow),
                               _c.c_int(ldao), _ptr(part[s]), _c.c_int(nao), _c.c_int(nao), _c.c_int(nao),
                               _c.c_long(ng16), _c.c_int(2), _c.c_int(nsplit_w), st)
        v = torch.empty((2, nao, nao), dtype=f64, device=dev)
        for s in range(2):
            self._call('reduce_sym', lib.PAMD_reduce_sym, _ptr(part[s]), _c.c_int(nsplit_max), _c.c_int(nao),
                       _c.c_int(nao), _ptr(v[s]), st)
        self._allreduce([v, acc], world)
        a = acc.cpu().numpy()
        return a[:2].copy(), float(a[2]), v.cpu().numpy()
