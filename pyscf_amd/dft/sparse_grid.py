"""Block-sparse grid plan for the XC contractions on the MI355X.

The reference screens its grid products with per-block shell lists: ``non0tab`` / ``screen_index`` from
``gen_grid.make_mask`` -> ``GTO_screen_index`` (pyscf/dft/gen_grid.py:455-484, pyscf/lib/gto/grid_ao_drv.c:32-123), consumed
by ``eval_gto`` and by ``VXCdot_ao_dm_sparse`` / ``VXCdot_ao_ao_sparse`` (pyscf/lib/dft/nr_numint_sparse.c:226-304,
:890-973; pyscf/dft/numint.py:836-957,2845).  Here the same information is turned into a data layout: the box-sorted grid is
cut into tiles of ``G`` consecutive points, every tile keeps the list of AO shells with a value (or gradient component)
above ``cutoff`` somewhere on it, and the AO values of exactly those functions are stored compacted,

    ao_c[tile] = [ncomp][G][ld_t]          ld_t = round_up(#active functions, 16),

so that every contraction of nr_rks / nr_uks is a dense FP64-MFMA GEMM on a compact operand (csrc/xc_sparse.hip).  The
compact image is small enough to stay resident in HBM across SCF iterations ((H2O)_32 cc-pVTZ, level-3 grid: 14 GB
against 64 GB dense), which also takes the AO evaluation out of the SCF loop.
"""
import ctypes

import numpy as np

from .. import lib as _lib_mod

_c = ctypes


def _ptr(t):
    return _c.c_void_p(t.data_ptr())


def _round_up(x, m):
    return (x + m - 1) // m * m


class SparsePlan:
    """Tile tables of one (molecule, grid, ncomp) on one rank.

    Attributes (device tensors unless noted):
      G, ntile_all            tile size, number of tiles of the whole grid (host ints)
      tiles                   host int array: global tile numbers owned by this rank (round-robin over ranks)
      ld, ao_off, aow_off, idx_off, idx      tables of xc_sparse.hip, one entry per LOCAL tile; ao_off / aow_off are
                              relative to the tile's chunk (chunk_ao_base adds the chunk's offset in the cached image)
      weights                 [ntile_local * G] quadrature weights in local tile order (zero padded)
      chunks                  host list of dicts: tile range [t0, t1), work list, ld_max, ao / aow sizes
      ao_c                    cached compact AO image (or None: recomputed chunk by chunk in every call)
      density                 host float: mean nsub / nao over the local tiles (reported by bench.py)
    """

    def __init__(self, ni, mol, grids, gga, dev, rank=0, world=1):
        import torch
        self.ni, self.mol, self.gga, self.dev = ni, mol, gga, dev
        self.ncomp = 4 if gga else 1
        self.G = G = int(ni.sparse_tile)
        assert G % 128 == 0
        sh = ni._shell_tables(mol, dev)
        self.nao, self.nsh = sh['nao'], sh['nsh']
        self.ldao = _round_up(self.nao, 16)
        self.coords_dev, wdev = ni._grid_tables(grids, dev)
        ngrids = self.ngrids = grids.size
        self.ntile_all = -(-ngrids // G)
        self.tiles = np.arange(rank, self.ntile_all, world)
        nloc = self.nloc = len(self.tiles)
        # weights in local tile order, zero padded
        wpad = torch.zeros(self.ntile_all * G, dtype=torch.float64, device=dev)
        wpad[:ngrids] = wdev
        self.weights = wpad.view(self.ntile_all, G)[torch.from_numpy(self.tiles).to(dev)].contiguous().view(-1)
        self._runs = self._tile_runs(max(1, int(ni.block_bytes // (self.ncomp * self.ldao * 8 * G))))
        active = self._screen(sh)
        self._tables(sh, active)
        # launch groups: as few as the orbital-product work space allows (ncomp x nocc_pad x points doubles per buffer, at most
        # `chunk_buffer_bytes` each; the response kernels hold three of them)
        nocc_hint = _round_up(max(getattr(mol, 'nelectron', 2) // 2, 1), 16)
        cap = int(getattr(ni, 'chunk_buffer_bytes', 6 << 30) // (self.ncomp * 8 * nocc_hint))
        self._chunks(max(G, min(int(ni.sparse_chunk_points), max(cap, 131072))))
        self.ao_c = None
        if not self._cache_fits() and self.max_chunk_points > 131072:
            # the compact image is recomputed chunk by chunk in every call: small launch groups bound that buffer (one big group
            # would need the whole image at once - 17.7 GiB at taxol size, found by the r04 taxol bench)
            self._chunks(131072)
        if self._cache_fits():
            self.ao_c = torch.empty(self.ao_total + 256, dtype=torch.float64, device=dev)
            self.ao_c[self.ao_total:].zero_()
            for ch in self.chunks:
                self.fill_chunk(ch, self.ao_c[ch['ao_base']:])
            from ..lib import hbm
            hbm.hold(self.dev, 'xc_image', (self.ao_total + 256) * 8)      # the XC share of the one HBM budget (lib/hbm.py)

    def __del__(self):
        try:
            if getattr(self, 'ao_c', None) is not None:
                from ..lib import hbm
                hbm.drop(self.dev, 'xc_image')
        except Exception:
            pass

    # -- geometry of the dense evaluation passes -------------------------------------------------------------------
    def _tile_runs(self, max_tiles):
        """[(first local tile, count)] runs of local tiles that are consecutive global tiles, at most max_tiles long."""
        runs, i = [], 0
        t = self.tiles
        while i < len(t):
            j = i + 1
            while j < len(t) and t[j] == t[j - 1] + 1 and j - i < max_tiles:
                j += 1
            runs.append((i, j - i))
            i = j
        return runs

    def _dense_block(self, i0, cnt, buf):
        """Dense AO values (PAMD_eval_ao) of local tiles [i0, i0 + cnt): returns (view [ncomp][rows][ldao], nvalid)."""
        G = self.G
        g0 = int(self.tiles[i0]) * G
        ng = min(cnt * G, self.ngrids - g0)
        self.ni.eval_ao_block(self.mol, self.coords_dev, g0, ng, self.gga, buf, buf.shape[1], self.ldao, None)
        return ng

    def _dense_buffer(self):
        import torch
        rows = max(c for _, c in self._runs) * self.G
        return torch.zeros((self.ncomp, rows, self.ldao), dtype=torch.float64, device=self.dev)

    def _screen(self, sh):
        """active[local tile][shell]: some AO value / gradient component of the shell exceeds the cutoff on the tile
        (value-based form of GTO_screen_index's exponent estimate)."""
        import torch
        G, nao, nsh = self.G, self.nao, self.nsh
        buf = self._dense_buffer()
        fn2sh = sh['fn2sh'].long()
        out = np.zeros((self.nloc, nsh), dtype=bool)
        for i0, cnt in self._runs:
            ng = self._dense_block(i0, cnt, buf)
            a = buf[:, :cnt * G, :nao].abs().amax(dim=0)
            if ng < cnt * G:
                a[ng:].zero_()
            fmax = a.view(cnt, G, nao).amax(dim=1)                           # [tile][function]
            smax = torch.zeros((cnt, nsh), dtype=torch.float64, device=self.dev)
            smax.scatter_reduce_(1, fn2sh[None, :].expand(cnt, nao), fmax, 'amax', include_self=True)
            out[i0:i0 + cnt] = (smax > self.ni.sparse_cutoff).cpu().numpy()
        self._dense = buf
        return out

    def _tables(self, sh, active):
        import torch
        nao, G = self.nao, self.G
        ao0 = sh['ao0'].cpu().numpy()
        nfn = 2 * sh['l'].cpu().numpy() + 1
        nsh = len(ao0)
        if nsh and np.array_equal(ao0, np.concatenate([[0], np.cumsum(nfn)[:-1]])):
            # shells in AO order (always, for this package's Mole): one boolean [tile][function] table instead of a Python loop over
            # the tiles (r06: 0.23 s -> 0.03 s of the plan at config 3); row-major nonzero() lists a tile's functions ascending,
            # exactly the concatenation of its active shells' ranges
            fmask = active[:, np.repeat(np.arange(nsh), nfn)]
            cnt = fmask.sum(axis=1)
            ld = (np.maximum(cnt, 1) + 15) // 16 * 16
            ld = ld.astype(np.int32)
            off = np.concatenate([[0], np.cumsum(ld)[:-1]]).astype(np.int64) if self.nloc else np.zeros(0, np.int64)
            idx_all = np.full(int(ld.sum()), nao, np.int32)
            rows, cols = np.nonzero(fmask)
            first = np.concatenate([[0], np.cumsum(cnt)[:-1]]) if self.nloc else np.zeros(0, np.int64)
            idx_all[off[rows] + (np.arange(len(rows)) - first[rows])] = cols
            self.nsub_host = cnt.astype(np.int64)
        else:
            ld = np.zeros(self.nloc, np.int32)
            idx_list = []
            for i in range(self.nloc):
                shells = np.nonzero(active[i])[0]
                fns = np.concatenate([np.arange(ao0[s], ao0[s] + nfn[s]) for s in shells]) if len(shells) else np.zeros(0, int)
                l = _round_up(max(len(fns), 1), 16)
                ld[i] = l
                row = np.full(l, nao, np.int32)
                row[:len(fns)] = fns
                idx_list.append(row)
            self.nsub_host = np.array([int((r < nao).sum()) for r in idx_list])
            idx_all = np.concatenate(idx_list) if self.nloc else np.zeros(0, np.int32)
        self.ld_host = ld
        self.density = float(self.nsub_host.mean() / nao) if self.nloc else 0.0
        self.density2 = float((self.nsub_host.astype(float) ** 2).mean() / nao ** 2) if self.nloc else 0.0
        self.idx_off_host = np.concatenate([[0], np.cumsum(ld)[:-1]]).astype(np.int64) if self.nloc else np.zeros(0, np.int64)
        self.idx = torch.from_numpy(idx_all).to(self.dev)
        self.ld = torch.from_numpy(ld).to(self.dev)
        self.idx_off = torch.from_numpy(self.idx_off_host).to(self.dev)

    def _chunks(self, max_points):
        import torch
        G, ncomp = self.G, self.ncomp
        per = max(1, int(max_points // G))
        self.chunks = []
        ao_off = np.zeros(self.nloc, np.int64)
        aow_off = np.zeros(self.nloc, np.int64)
        base = 0
        for t0 in range(0, self.nloc, per):
            t1 = min(t0 + per, self.nloc)
            ldc = self.ld_host[t0:t1].astype(np.int64)
            ao_sz = ncomp * G * ldc
            aow_sz = G * ldc
            ao_off[t0:t1] = np.concatenate([[0], np.cumsum(ao_sz)[:-1]])
            aow_off[t0:t1] = np.concatenate([[0], np.cumsum(aow_sz)[:-1]])
            # work items {tile (relative to the chunk), tm, tn}, largest tiles first
            work = []
            for t in np.argsort(-ldc, kind='stable'):
                nt = -(-int(ldc[t]) // 128)
                for tm in range(nt):
                    for tn in range(nt):
                        work.append((t, tm, tn))
            # r04: piece pairs (i >= j) of a balanced cut of every tile's 16-column groups (PAMD_sub_vmat_sym)
            ld32 = np.ascontiguousarray(ldc, dtype=np.int32)
            lib = _lib_mod.load_library()
            lib.PAMD_sub_vmat_work.restype = _c.c_long
            nsym = int(lib.PAMD_sub_vmat_work(ld32.ctypes.data_as(_c.c_void_p), _c.c_int(len(ld32)), _c.c_void_p(0)))
            wsym = np.zeros(max(nsym, 1) * 6, np.int32)
            lib.PAMD_sub_vmat_work(ld32.ctypes.data_as(_c.c_void_p), _c.c_int(len(ld32)), wsym.ctypes.data_as(_c.c_void_p))
            self.chunks.append(dict(t0=t0, t1=t1, ao_base=base, ao_size=int(ao_sz.sum()), aow_size=int(aow_sz.sum()),
                                    ld_max=int(ldc.max()), nwork=len(work),
                                    work=torch.from_numpy(np.asarray(work, np.int32).reshape(-1)).to(self.dev),
                                    nwork_sym=nsym, work_sym=torch.from_numpy(wsym).to(self.dev)))
            base += int(ao_sz.sum())
        self.ao_total = base
        self.ao_off = torch.from_numpy(ao_off).to(self.dev)
        self.aow_off = torch.from_numpy(aow_off).to(self.dev)
        self.max_ao_chunk = max([c['ao_size'] for c in self.chunks] + [0])
        self.max_aow_chunk = max([c['aow_size'] for c in self.chunks] + [0])
        self.max_chunk_points = max([(c['t1'] - c['t0']) * self.G for c in self.chunks] + [0])

    def _cache_fits(self):
        import torch
        mode = self.ni.ao_cache
        if mode is False or self.nloc == 0:
            return False
        if mode is True:
            return True
        free = torch.cuda.mem_get_info(self.dev)[0] + torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
        return (self.ao_total + 256) * 8 + self.ni.ao_cache_reserve <= free

    # -- per-chunk device pointers ------------------------------------------------------------------------------------
    def fill_chunk(self, ch, out):
        """Compact AO values of one chunk into `out` (a 1-D view that starts at the chunk's base)."""
        import torch
        lib = _lib_mod.load_library()
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        buf = self._dense if getattr(self, '_dense', None) is not None else self._dense_buffer()
        self._dense = buf
        for i0, cnt in self._runs:
            a, b = max(i0, ch['t0']), min(i0 + cnt, ch['t1'])
            if a >= b:
                continue
            ng = self._dense_block(a, b - a, buf)
            self.ni._call('sub_gather', lib.PAMD_sub_gather_ao, _ptr(buf), _c.c_long(buf.shape[1]), _c.c_int(self.ldao),
                          _c.c_int(self.ncomp), _c.c_long(0), _c.c_long(ng), _ptr(self.ao_off[a:]), _ptr(self.idx_off[a:]),
                          _ptr(self.ld[a:]), _ptr(self.idx), _c.c_int(b - a), _c.c_int(self.G),
                          _c.c_int(int(self.ld_host[a:b].max())), _c.c_int(self.nao), _ptr(out), st)

    def release_dense(self):
        self._dense = None
