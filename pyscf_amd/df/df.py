"""``DF`` object: owner of the Cholesky-decomposed 3-centre tensor on the GPU.

Duck-types ``pyscf.df.df.DF`` (pyscf/df/df.py:40-337): ``build`` (:147-199), ``reset``
(:204-212), ``loop`` (:214-242), ``get_naoaux`` (:248-257), ``get_jk`` (:259-267), so that
``mf.with_df = pyscf_amd.df.DF(mol, auxbasis)`` (or ``mf.density_fit(with_df=...)``,
pyscf/df/df_jk.py:31,77-105) routes ``_DFHF.get_jk`` (df_jk.py:150-179) onto the MI355X path.

The tensor lives in HBM as the row shard ``cderi[l0:l1, :nao_pair]`` of this rank
(aux-index sharding); with one process it is the whole ``_cderi``.
"""
import os

import numpy as np

from . import addons, df_jk


def _mol_nao(mol):
    """AO count of any molecule-like object: a Mole of this package or of stock PySCF, or a bare holder of libcint tables
    (spherical functions: sum over shells of (2 l + 1) nctr, pyscf/gto/mole.py ao_loc_nr)."""
    if hasattr(mol, 'nao_nr'):
        return int(mol.nao_nr())
    if hasattr(mol, 'nao'):
        return int(mol.nao)
    bas = np.asarray(mol._bas)
    return int(((2 * bas[:, 1] + 1) * bas[:, 3]).sum())


class DF:
    blockdim = 240     # pyscf/df/df.py:95

    def __init__(self, mol, auxbasis=None, device=None, group=None):
        self.mol = mol
        self.stdout = getattr(mol, 'stdout', None)
        self.verbose = getattr(mol, 'verbose', 0)
        self.max_memory = getattr(mol, 'max_memory', 4000)
        self._auxbasis = auxbasis
        self.auxmol = None
        self._cderi = None          # optional host ndarray (naux, nao_pair) to upload
        self._packed = None         # torch CUDA tensor [rows l0..l1][nao_pair]: the reference's packed layout (`_cderi_dev` property)
        # r06: ONE resident copy of the tensor.  'square': sq[L][rows][rows] (both triangles, what the K half
        # transform streams by LDS-DMA) INSTEAD of the packed rows - 2x the packed bytes, not packed + image = 3x; the J passes
        # read the p >= q runs of the square rows, loop / save / export pack on the fly.  'packed': the reference's rows (+ the
        # optional partial image / diagonal-block image of r03) when 2x does not fit.  'auto': square when the budget allows.
        self.layout = 'auto'
        self.prefer_image = True    # 'auto': packed rows + a full image while all three copies fit the budget (the fastest layout
                                    # beside the co-running SYRK, measured); False: the square rows whenever 2x fits
        self._layout = None         # what build() decided
        self.xc_image_hint = 0      # bytes the XC leg of the same calculation will cache in HBM (KS objects set it): part of the budget
        self._cderi_to_save = None
        self._naux = None
        self.device = device
        self.group = group
        # HBM budget for the half-transformed block X (MI355X: 288 GB per GPU)
        self.k_block_bytes = 12 << 30      # X block: config 3 in ONE block (10.6 GB; one kernel boundary less per step, -0.5 %)
        self.j2_policy = 'auto'    # second J pass 'overlap' (side stream, beside a plain SYRK) | 'serial' (in line, re-tiled SYRK) | 'fused' (r05: inside the SYRK kernel) |
                                   # 'auto': both timed once per shape (df_jk.get_jk_device)
        self.j2_tune = 'lazy'      # how 'auto' times them: 'lazy' - on the caller's own calls, one candidate per call, no extra builds
                                   # (an SCF pays nothing); 'eager' - trial builds before the first answer (bench.py, kernel tools)
        self.j2_lazy_reps = 2      # samples per candidate before 'lazy' settles
        self.j2_tune_min_bytes = 4 << 30
        self.j2_try_fused = True   # 'auto' also times the pass INSIDE the SYRK kernel (PAMD_syrk_jfused, r05) and takes it when >= 1 % faster
        self.k_e2_pipeline = 1     # sub-blocks of a K block whose half transforms are queued back to back (df_jk._vk_mo)
        self.k_nsplit = None       # k-splits of the K = X^T X product; None: df_jk.syrk_plan picks tile shape and splits
        self.k_syrk_reserve = 16   # > 0: beside a co-running J pass 2 the balanced re-tiled SYRK, sized to leave that many of the
                                   # 512 workgroup slots to the pass, instead of the plain grid (df_jk._vk_mo)
        self.lindep = 1e-7         # pyscf/df/incore.py:33
        self.decompose_j2c = 'CD'  # 'ED': eigen-decompose the metric even when it is positive definite (df/grad/rhf.py:45)
        self.omega = 0.0           # > 0: long-range, < 0: short-range tensor (set by range_coulomb)
        self._rsh_df = {}
        self.incore_anyway = False  # mol.incore_anyway analogue (df_jk.py:282): force the tensor path
        self._eng = None
        self.overlap_jk = True     # run J (HBM-bound) on a second stream beside K (MFMA-bound)
        # K path on an unpacked (square) second copy of the tensor, 2x the packed size, when HBM allows ('auto'),
        # always (True) or never (False): plain row panels stream by LDS-DMA, no symmetric unpack in the hot loop
        self.k_square = 'auto'
        self.k_square_reserve = 48 << 30     # HBM left free after the copy (X block, partial K, XC blocks, ...)
        self._cderi_sq = None
        self._sq_nao = 0
        self.k_diag = True                  # packed-operand rows: keep the diagonal 128 x 128 blocks unpacked (diag_image)
        self._cderi_diag = None
        self._diag_row0 = 0
        self.fuse_j_pass1 = True            # MO branch: first J pass from the half transform's epilogue
        self.overlap_split = True           # two-pass J: pass 1 / pass 2 behind the first / second SYRK block
        self.factorize_hermitian_dm = True  # hermi=1 DMs without orbitals: eigen-factorise, use the MO kernels
        self.kernel_timer = None   # df_jk.KernelTimer() to collect per-kernel HIP-event timings
        self.outcore = True        # a tensor that does not fit the device goes to the C handle (HBM + streamed host rows) in build()
        self.outcore_device_bytes = 0   # > 0: cap on the device memory the tensor may take (tests, memory-constrained callers)
        self._native = None
        self._ws = {}

    # -- distributed geometry ----------------------------------------------------------
    @property
    def world_size(self):
        if getattr(self, '_shard_override', None) is not None:     # (rank, world) without a process group:
            return self._shard_override[1]                         # build / contract one shard only (tools, tests)
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_world_size(self.group)
        except ImportError:
            pass
        return 1

    @property
    def rank(self):
        if getattr(self, '_shard_override', None) is not None:
            return self._shard_override[0]
        if self.world_size > 1:
            import torch.distributed as dist
            return dist.get_rank(self.group)
        return 0

    # -- the tensor, whichever layout holds it -------------------------------------------------------
    @property
    def _cderi_dev(self):
        """The PACKED rows (naux_local, nao_pair) - the reference's `_cderi` layout (pyscf/df/df.py:59-72).  In the square layout
        they do not exist until somebody asks: the first access packs a copy out of the square rows (gradients, get_eri / ao2mo,
        tests; the J/K path never does).  Internal code tests `has_tensor()` / `_packed`, never this property against None."""
        if self._packed is None and self._cderi_sq is not None and self._layout == 'square':
            self._packed = self.packed_rows(0, self._cderi_sq.shape[0])
        return self._packed

    @_cderi_dev.setter
    def _cderi_dev(self, value):
        self._packed = value
        if value is not None and self._layout is None:
            self._layout = 'packed'                  # a caller-provided packed tensor (tools, tests): the r03-r05 path

    SQ_STRIDE_PAD = 0            # doubles added to rows * ld between consecutive aux rows of the square layout.  rows * ld * 8 is a
                                 # multiple of 32 KB at nao = 1856 / 2228 and of 8 MB at nao = 3072, and the same (p, q) of consecutive
                                 # aux rows is what a J pass has in flight - an HBM channel conflict was the suspicion when the square
                                 # J pass measured 5.2 instead of 6.6 TB/s.  Measured with 288 (profiles/r06/kbench_square_layout.log):
                                 # no difference (11.33 vs 11.22 ms alone, 111.9-114.3 vs 111.4-112.4 ms per step) - so no pad; the
                                 # stride stays an explicit argument of every kernel that walks the square rows

    @classmethod
    def alloc_square(cls, nL, rows, device):
        """Zeroed square rows sq[nL][rows][rows] with the padded aux-row stride (a strided view of one flat buffer, 256 doubles of
        slack behind the last row for the LDS-DMA kernels' whole-panel reads)."""
        import torch
        ls = rows * rows + cls.SQ_STRIDE_PAD
        buf = torch.zeros(max(nL, 1) * ls + 256, dtype=torch.float64, device=device)
        return buf.as_strided((nL, rows, rows), (ls, rows, 1))

    def has_tensor(self):
        return self._packed is not None or (self._layout == 'square' and self._cderi_sq is not None)

    def tensor_shape(self):
        """(local aux rows, nao_pair) without materialising anything."""
        if self._packed is not None:
            return tuple(self._packed.shape)
        sq = self._cderi_sq
        nao = self._sq_nao
        return sq.shape[0], nao * (nao + 1) // 2

    def tensor_device(self):
        return (self._packed if self._packed is not None else self._cderi_sq).device

    def packed_rows(self, b0, b1, out=None):
        """Device tensor (b1 - b0, nao_pair) of the packed rows [b0, b1) of this rank: a view in the packed layout, a fresh
        pack (PAMD_pack_tril_rows) out of the square layout."""
        import torch
        import ctypes as _c
        from .. import lib as _lib
        if self._packed is not None:
            return self._packed[b0:b1]
        sq = self._cderi_sq
        nao = self._sq_nao
        npair = nao * (nao + 1) // 2
        if out is None:
            out = torch.empty((b1 - b0, npair), dtype=torch.float64, device=sq.device)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        if b1 > b0:
            _lib.check(_lib.load_library().PAMD_pack_tril_rows(_c.c_void_p(sq[b0:b1].data_ptr()), _c.c_long(sq.stride(0)),
                                                               _c.c_int(sq.shape[2]), _c.c_int(nao), _c.c_int(b1 - b0),
                                                               _c.c_void_p(out.data_ptr()), st))
        return out[:b1 - b0]

    def to_packed_layout(self, release=True):
        """The consumers that need the packed tensor AND the memory of the square one (analytic gradients: W has the tensor's
        size): pack a copy when both fit, else drop the square rows and rebuild the tensor packed from the integrals."""
        import torch
        if self._layout != 'square' or not self.has_tensor():
            return self
        dev = self.tensor_device()
        nL, npair = self.tensor_shape()
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        if self._packed is None and nL * npair * 8 + (8 << 30) <= free:
            self._packed = self.packed_rows(0, nL)
        if self._packed is not None:
            self._cderi_sq = None
            self._layout = 'packed'
            if release:
                torch.cuda.empty_cache()
            return self
        mol_ok = self.mol is not None and not isinstance(self._cderi, (str, np.ndarray))
        if not mol_ok:
            raise MemoryError('DF.to_packed_layout: no room for a packed copy beside the square tensor and no integrals to rebuild from')
        self._cderi_sq = None
        self._ws = {}
        torch.cuda.empty_cache()
        self._layout = None
        self.layout, self.k_square = 'packed', False
        return self.build()

    @staticmethod
    def shard_range(naux, rank, world):
        """Contiguous, row-count-balanced aux range of `rank`."""
        base, rem = divmod(naux, world)
        l0 = rank * base + min(rank, rem)
        return l0, l0 + base + (1 if rank < rem else 0)

    def _side_stream(self):
        import torch
        if getattr(self, '_side', None) is None:
            # (k_side_priority = 1: the lowest queue priority for the second J pass's stream - what cured a bimodal SYRK in the C
            # handle, csrc/df_handle.hip; measured neutral here, profiles/r06/README.md: default 0)
            self._side = torch.cuda.Stream(device=self.tensor_device(), priority=int(getattr(self, 'k_side_priority', 0)))
        return self._side

    # -- integral-direct J support (no tensor) ---------------------------------------------------
    def _direct_engine(self):
        """(IntEngine, lower Cholesky factor of j2c) cached for df_jk.get_j."""
        if getattr(self, '_eng', None) is None:
            import scipy.linalg
            from ..gto.moleintor import get_engine
            if self.auxmol is None:
                self.auxmol = addons.make_auxmol(self.mol, self.auxbasis)
            self._eng = get_engine(self.mol, self.auxmol, self._device(), self.omega)
            j2c = self._eng.int2c2e().cpu().numpy()
            self._j2c_low = scipy.linalg.cholesky((j2c + j2c.T) * .5, lower=True)
            self._naux = self._eng.aux.nao
        return self._eng, self._j2c_low

    def _direct_slabs(self, eng, slab_bytes=8 << 30):
        naux = eng.aux.nao
        max_rows = max(int(slab_bytes // (naux * 8)), 1)
        slabs, sh0, nsh = [], 0, eng.ao.n
        while sh0 < nsh:
            sh1 = sh0 + 1
            while sh1 < nsh and eng.slab_rows(sh0, sh1 + 1)[1] - eng.slab_rows(sh0, sh1 + 1)[0] <= max_rows:
                sh1 += 1
            slabs.append((sh0, sh1))
            sh0 = sh1
        return slabs

    def _direct_buffer(self, rows, naux, dev):
        import torch
        buf = self._ws.get('T')
        if buf is None or buf.numel() < rows * naux or buf.device != dev:
            buf = torch.empty(rows * naux, dtype=torch.float64, device=dev)
            self._ws['T'] = buf
        return buf[:rows * naux].view(rows, naux)

    def square_image(self):
        """sq[L][q][p] (rows and ld rounded up to 16, zero padded) of the first `sq.shape[0]` of this rank's cderi rows, or None.
        k_square = 'auto': all rows when HBM allows (k_square_reserve stays free); else as many rows as fit (r03: taxol on one
        GPU - 111 GB packed, the 222 GB image does not fit beside it, the image of 60 % of the rows does; their half transform
        runs on the LDS-DMA square kernel at 73 TF/s instead of 54 on the packed-operand one); fewer than 1/4 of the rows: none."""
        import torch
        import ctypes as _c
        from .. import lib as _lib
        if self._cderi_sq is not None or self.k_square is False or self._packed is None:
            return self._cderi_sq
        naux, npair = self._packed.shape
        nao = int((np.sqrt(8.0 * npair + 1) - 1) / 2 + .5)
        rows = (nao + 15) // 16 * 16
        if naux == 0 or nao * (nao + 1) // 2 != npair:
            return None
        nsq = naux
        if self.k_square == 'auto':
            free = torch.cuda.mem_get_info(self._packed.device)[0] + torch.cuda.memory_reserved(self._packed.device) \
                - torch.cuda.memory_allocated(self._packed.device)
            # the optional image takes what the budget leaves: the reserve AND what the XC leg will cache (never a K image that
            # evicts the AO cache)
            from ..lib import hbm
            room = free - self.k_square_reserve - max(0, int(self.xc_image_hint) - hbm.held(self._packed.device, 'xc_image')) - 256 * 8
            nsq = min(naux, int(room // (rows * rows * 8))) if room > 0 else 0
            if getattr(self, 'k_square_max_rows', None) is not None:       # cap (tests, tuning)
                nsq = min(nsq, int(self.k_square_max_rows))
            if nsq < naux:
                nsq = nsq // 64 * 64                       # whole K blocks are cut at multiples of 64 rows (df_jk._vk_mo)
            if nsq < max(naux // 4, 1) or not getattr(self, 'k_square_partial', True) and nsq < naux:
                self.k_square = False
                return None
        so = _lib.load_library()
        # unpack_tril writes every [p < nao][q < rows] entry; only the pad rows p >= nao (and the slack) need zeroing
        buf = torch.empty(nsq * rows * rows + 256, dtype=torch.float64, device=self._packed.device)
        buf[nsq * rows * rows:].zero_()
        if rows > nao:
            buf[:nsq * rows * rows].view(nsq, rows, rows)[:, nao:, :].zero_()
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(so.PAMD_unpack_tril(_c.c_void_p(self._packed.data_ptr()), _c.c_long(npair), _c.c_int(nsq),
                                       _c.c_int(nao), _c.c_void_p(buf.data_ptr()), _c.c_int(rows), _c.c_int(rows), st))
        self._cderi_sq = buf[:nsq * rows * rows].view(nsq, rows, rows)
        self._sq_nao = nao
        return self._cderi_sq

    def diag_image(self):
        """(dg, row0): dg[L - row0][P][128][128] = both triangles of the 128 x 128 blocks on the diagonal of B_L for the cderi
        rows L >= row0 that the square image does NOT cover (all rows when there is none), or (None, 0).  14 % of the packed
        size at nao 1856 (8.7 GB at BASELINE config 3); with it the packed-operand half transform reads the k-tiles that cross
        the diagonal once and unmasked (PAMD_nr_e2_symm_diag, r03).  k_diag = False switches it off."""
        import torch
        import ctypes as _c
        from .. import lib as _lib
        if self._cderi_diag is not None or not getattr(self, 'k_diag', True) or self._packed is None:
            return self._cderi_diag, self._diag_row0
        naux, npair = self._packed.shape
        nao = int((np.sqrt(8.0 * npair + 1) - 1) / 2 + .5)
        if naux == 0 or nao * (nao + 1) // 2 != npair or nao < 128:
            return None, 0
        sq = self.square_image()
        row0 = sq.shape[0] if sq is not None else 0
        if row0 >= naux:
            return None, 0
        ldx = (nao + 15) // 16 * 16
        so = _lib.load_library()
        so.PAMD_e2_diag_size.restype = _c.c_long
        n = so.PAMD_e2_diag_size(_c.c_int(naux - row0), _c.c_int(ldx))
        dev = self._packed.device
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        if free - (4 << 30) < n * 8:       # (the square image has left k_square_reserve free; the work buffers live in there)
            self.k_diag = False
            return None, 0
        buf = torch.empty(n, dtype=torch.float64, device=dev)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(so.PAMD_e2_diag_blocks(_c.c_void_p(self._packed[row0:].data_ptr()), _c.c_long(npair),
                                          _c.c_int(naux - row0), _c.c_int(nao), _c.c_int(ldx), _c.c_void_p(buf.data_ptr()), st))
        self._cderi_diag = buf.view(naux - row0, -1)
        self._diag_row0 = row0
        return self._cderi_diag, row0

    def drop_square_image(self, release=False):
        """Give the HBM of the square copy back (the gradient path needs it for W and Z)."""
        import torch
        if self._layout == 'square':
            self.to_packed_layout(release)      # the square rows ARE the tensor: pack them (or rebuild packed) first
        self._cderi_sq = None
        self._cderi_diag = None
        self._diag_row0 = 0
        # r06: the block stays in torch's caching allocator - W and the Z slabs are carved out of it.  Handing it back to the driver
        # (empty_cache) and asking again costs a VRAM clear of ~38 ms per GB on this stack (tools/probe/alloc_after_exit.py: 123 GB of
        # never-used HBM in 0.48 s, the same 123 GB after a free in 4.7 s): the gradient's 61 GB W then took 3.6 s instead of 0.39 s
        # whenever the driver handed it the pages just released (profiles/r06/grad_h2o32_rhf_dirty_pages.json)
        if release:
            torch.cuda.empty_cache()

    def _workspace(self, name, shape):
        """Persistent HBM scratch (no per-iteration hipMalloc): returns a view of `shape`."""
        import torch
        n = int(np.prod(shape))
        buf = self._ws.get(name)
        if buf is None or buf.numel() < n or buf.device != self.tensor_device():
            # +256 doubles of slack: the LDS-DMA GEMM reads whole 128-column panel rows
            buf = torch.zeros(n + 256, dtype=torch.float64, device=self.tensor_device())
            self._ws[name] = buf
        return buf[:n].view(*shape)

    # -- reference attributes -------------------------------------------------------------
    @property
    def auxbasis(self):
        return self._auxbasis

    @auxbasis.setter
    def auxbasis(self, x):
        if self._auxbasis != x:
            self.reset()
            self._auxbasis = x

    def reset(self, mol=None):
        if mol is not None:
            self.mol = mol
        self.auxmol = None
        self._cderi = None
        self._packed = None
        self._layout = None
        self._cderi_sq = None
        self._cderi_diag = None
        self._diag_row0 = 0
        self._naux = None
        if getattr(self, '_native', None) is not None:
            self._native.reset()
        self._native = None
        self._ws = {}
        self._eng = None
        self._rsh_df = {}          # pyscf/df/df.py:210: the range-separated children hold the old mol / auxmol
        self._side = None
        self.__dict__.pop('_j2_lazy', None)            # half-finished schedule trials belong to the old tensor
        self.__dict__.pop('_j2_policy_cache', None)
        return self

    def _device(self):
        import torch
        if self.device is not None:
            return torch.device(self.device)
        if not torch.cuda.is_available():
            raise RuntimeError('pyscf_amd.df.DF needs a HIP device (MI355X); none is visible '
                               'and there is no CPU fallback for the DF J/K path')
        return torch.device('cuda', torch.cuda.current_device())

    def build(self):
        import torch
        if getattr(self, '_native', None) is not None:
            return self
        dev = self._device()
        if isinstance(self._cderi, str):
            from ..lib import hdf5
            if hdf5.is_hdf5(self._cderi):
                # the reference's own format: dataset 'j3c' (naux, nao_pair) (pyscf/df/df.py:97-99, outcore.py:217-221);
                # every rank reads exactly its aux rows
                with hdf5.File(self._cderi) as f:
                    # one dataset 'j3c' (naux, nao_pair), or the group 'j3c/0..N' of column blocks that stock PySCF's
                    # outcore.cholesky_eri_b writes by default (_compatible_format = False; df.py:227-241)
                    blocks = f.column_blocks('j3c') or [f['j3c']]
                    naux = blocks[0].shape[0]
                    ncol = sum(b.shape[1] for b in blocks)
                    nao = _mol_nao(self.mol)
                    if any(len(b.shape) != 2 or b.shape[0] != naux for b in blocks) or ncol != nao * (nao + 1) // 2:
                        raise RuntimeError("%s: 'j3c' holds %d columns, expected nao_pair = %d for nao = %d (s2-packed (naux, "
                                           "nao_pair) tensor, pyscf/df/df.py:59-72)" % (self._cderi, ncol, nao * (nao + 1) // 2, nao))
                    l0, l1 = self.shard_range(naux, self.rank, self.world_size)
                    if (not self._all_ranks_agree(not self._rows_do_not_fit(l1 - l0, ncol, dev))) and getattr(self, 'outcore', True):
                        # r05: the file's rows of this rank do not fit the device - they are NOT loaded: the 'j3c' dataset is
                        # mapped (np.memmap at H5Dget_offset) and the C handle streams the non-resident rows straight out of the
                        # mapping under the kernels in every build (pyscf/df/df.py:214-242: the reference reads its file block by
                        # block in DF.loop every iteration as well)
                        off = blocks[0].file_offset() if len(blocks) == 1 else None
                        if off is not None and not blocks[0].is_native_f64_le():
                            #  the mapping below reads raw bytes: anything but little-endian float64 would stream garbage
                            raise MemoryError("%s: the tensor does not fit the device and 'j3c' is not stored as little-endian "
                                              "float64 (the raw mapping cannot convert): rewrite it with DF.save()" % self._cderi)
                        if off is None:
                            raise MemoryError("%s: the tensor does not fit the device and 'j3c' is not one contiguous dataset "
                                              "(column-block group or chunked layout): rewrite it with DF.save()" % self._cderi)
                        mm = np.memmap(self._cderi, dtype='<f8', mode='r', offset=off, shape=(naux, ncol))
                        self._native_from_rows(mm[l0:l1], (l0, l1), naux, dev)
                        return self
                    self._cderi_dev = torch.empty((l1 - l0, ncol), dtype=torch.float64, device=dev)
                    c0 = 0
                    for d in blocks:
                        step = max(1, (1 << 30) // (d.shape[1] * 8))
                        for r0 in range(l0, l1, step):
                            r1 = min(r0 + step, l1)
                            self._cderi_dev[r0 - l0:r1 - l0, c0:c0 + d.shape[1]] = torch.from_numpy(d.read_rows(r0, r1)).to(dev)
                        c0 += d.shape[1]
                self._naux = naux
                return self._maybe_square()
            shard = self._shard_path(self._cderi)
            if self.world_size > 1 and os.path.exists(shard):
                # written by save() of a run with the same world size: this rank's rows, no re-sharding
                with np.load(shard) as z:
                    rows, l0, l1, naux = z['j3c'], int(z['l0']), int(z['l1']), int(z['naux'])
                if (l0, l1) != self.shard_range(naux, self.rank, self.world_size):
                    raise RuntimeError('%s holds aux rows [%d, %d) of %d: not the shard of rank %d of %d'
                                       % (shard, l0, l1, naux, self.rank, self.world_size))
                self._cderi_dev = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
                self._naux = naux
                return self._maybe_square()
            with open(self._cderi, 'rb') as f:        # one file = the FULL tensor (written by a single-rank run)
                self._cderi = np.load(f)
        if self._cderi is not None and isinstance(self._cderi, np.ndarray):
            # pre-computed FULL tensor handed over by the caller (pyscf/df/df.py:153-155); each rank keeps its rows
            naux = self._cderi.shape[0]
            l0, l1 = self.shard_range(naux, self.rank, self.world_size)
            if (not self._all_ranks_agree(not self._rows_do_not_fit(l1 - l0, self._cderi.shape[1], dev))) and getattr(self, 'outcore', True):
                # streamed out of the caller's array (C order, float64: as it is; anything else through one host copy of the rows)
                self._native_from_rows(np.ascontiguousarray(self._cderi[l0:l1], dtype=np.float64), (l0, l1), naux, dev)
                return self
            self._cderi_dev = torch.from_numpy(np.ascontiguousarray(self._cderi[l0:l1])).to(dev)
            self._naux = naux
            return self._maybe_square()
        if self.auxmol is None:
            self.auxmol = addons.make_auxmol(self.mol, self.auxbasis)
        from . import incore
        self._naux = self.auxmol.nao_nr()
        l0, l1 = self.shard_range(self._naux, self.rank, self.world_size)
        try:
            # the fit check comes BEFORE any metric work (the metric used to be computed and factorised twice, once by
            # cholesky_eri_gpu ahead of its own memory check and once more inside the C handle)
            if not self._all_ranks_agree(self.would_fit()):
                raise MemoryError('DF tensor shard of %d x %d doubles does not fit %s' % (
                    l1 - l0, _mol_nao(self.mol) * (_mol_nao(self.mol) + 1) // 2,
                    'DF.outcore_device_bytes' if self.outcore_device_bytes else 'the free device memory'))
            lay = self._choose_layout(l1 - l0, dev)
            t = incore.cholesky_eri_gpu(self.mol, self.auxmol, dev, l0, l1, lindep=self.lindep, omega=self.omega,
                                        decompose_j2c=self.decompose_j2c, layout=lay,
                                        slab_bytes=(12 << 30) if lay == 'square' else (24 << 30))
            if lay == 'square':
                self._cderi_sq, self._sq_nao, self._layout, self._packed = t, _mol_nao(self.mol), 'square', None
            else:
                self._packed, self._layout = t, 'packed'
            torch.cuda.empty_cache()          # the build's slab work space (<= 2 x 24 GB) back to the device: the budget's next clients see it
        except MemoryError:
            # The out-of-core twin (pyscf/df/outcore.py:109-232, pyscf/df/df.py:167): the tensor (r05: or this RANK's shard of it)
            # does not fit this device.  The object hands its rows to the C handle, which keeps what fits in HBM, the rest in
            # page-locked host memory, and streams it under the kernels in every build (PCIe-bound for those rows: slower, not
            # wrong).  J/K, loop, save and range_coulomb go through it - a rank of a multi-process job gets its shard's PARTIAL
            # J/K from the handle and all-reduces them like the in-core path; the HBM-resident SCF loop and the gradients need
            # the in-core tensor.
            if not getattr(self, 'outcore', True) or getattr(self, 'decompose_j2c', 'CD') != 'CD':
                raise
            from .native import NativeDF
            import torch as _torch
            idx = dev.index if dev.index is not None else _torch.cuda.current_device()
            _torch.cuda.empty_cache()                       # give the handle's hipMalloc the memory torch had cached
            sharded = self.world_size > 1 or getattr(self, '_shard_override', None) is not None
            self._native = NativeDF(self.mol, self.auxbasis, auxmol=self.auxmol, device=idx, lindep=self.lindep, omega=self.omega,
                                    max_device_bytes=int(getattr(self, 'outcore_device_bytes', 0)),
                                    shard=(self.rank, self.world_size) if sharded else None).build()
            self._naux = self._native.get_naoaux()
        if isinstance(self._cderi_to_save, str):
            self.save(self._cderi_to_save)
        return self
    kernel = build

    def _reserve_after_build(self, nL, rows):
        """HBM the rest of the calculation needs once the tensor is resident - the single budget of VERDICT r05 item 1, in the
        order tensor, X block, XC compact image, work space: the half-transformed block (DF.k_block_bytes; at most nL nao rows),
        split-K partials / J vectors / orbitals (4 GB), and - when the calculation has an XC leg (`xc_image_hint`, set by the
        Kohn-Sham classes) - the compact AO image plus the XC work buffers (12 GB)."""
        from ..lib import hbm
        r = min(int(self.k_block_bytes), nL * rows * rows * 8) + (4 << 30)
        if self.xc_image_hint:
            # (an SCF builds its XC plan BEFORE the tensor: what the plan already holds - image and work buffers - is not asked for twice)
            held = hbm.held(self._device(), 'xc_image')
            r += max(0, int(self.xc_image_hint) - held) + ((4 << 30) if held else (12 << 30))
        return r

    def _maybe_square(self):
        """Uploaded packed rows (a caller's array, a `_cderi` file) -> the square layout when DF.layout asks for it or ('auto')
        the budget allows: unpacked once (PAMD_unpack_tril), then the packed rows are released - 2x, not 3x."""
        import torch
        import ctypes as _c
        from .. import lib as _lib
        from ..lib import hbm
        if self._layout != 'packed' or self._packed is None or self.layout == 'packed' or self.k_square is False:
            return self
        nL, npair = self._packed.shape
        nao = int((np.sqrt(8.0 * npair + 1) - 1) / 2 + .5)
        if nL == 0 or nao * (nao + 1) // 2 != npair:
            return self
        rows = (nao + 15) // 16 * 16
        dev = self._packed.device
        if self.layout != 'square':
            free = hbm.free_bytes(dev)
            # (the packed rows are resident already: a full image beside them is preferred where it fits - see _choose_layout)
            luxury = nL * rows * rows * 8 + max(self._reserve_after_build(nL, rows), int(self.k_square_reserve))
            if getattr(self, 'prefer_image', True) and self._all_ranks_agree(luxury + (2 << 30) <= free):
                return self
            fits = nao >= 128 and nL * rows * rows * 8 + max(self._reserve_after_build(nL, rows), 4 << 30) <= free
            if not self._all_ranks_agree(fits):
                return self
        sq = self.alloc_square(nL, rows, dev)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        step = max(1, (4 << 30) // (npair * 8))
        for b0 in range(0, nL, step):             # (PAMD_unpack_tril_slab with the whole triangle as one slab: explicit aux-row stride)
            nb = min(step, nL - b0)
            _lib.check(_lib.load_library().PAMD_unpack_tril_slab(_c.c_void_p(self._packed[b0:].data_ptr()), _c.c_long(npair), _c.c_int(nb),
                                                                 _c.c_int(0), _c.c_int(nao), _c.c_void_p(sq[b0:].data_ptr()), _c.c_int(rows),
                                                                 _c.c_long(sq.stride(0)), st))
        torch.cuda.current_stream().synchronize()
        self._cderi_sq, self._sq_nao, self._layout, self._packed = sq, nao, 'square', None
        self._cderi_diag, self._diag_row0 = None, 0
        torch.cuda.empty_cache()
        return self

    def _choose_layout(self, nL, dev):
        """'square' when 2x the packed shard + the build's slab work space + the reserve fit the device (or DF.layout says
        so), else 'packed'.  Collective over the ranks like the out-of-core decision: one layout per job."""
        import torch
        if self.layout in ('packed', 'square'):
            return self.layout
        if self.k_square is False or dev.type != 'cuda':
            return 'packed'
        nao = _mol_nao(self.mol)
        if nao < 128:
            return 'packed'                       # tiny tensors: the exact-tile kernels read the packed rows directly
        rows = (nao + 15) // 16 * 16
        naux = self.auxmol.nao_nr()
        npair = nao * (nao + 1) // 2
        from ..lib import hbm
        free = hbm.free_bytes(dev)
        slab = min(12 << 30, npair * naux * 8)
        build_need = nL * rows * rows * 8 + slab + min(slab, slab * max(nL, 1) // max(naux, 1) + (1 << 20)) + (2 << 30)
        after_need = nL * rows * rows * 8 + self._reserve_after_build(nL, rows)
        # The order of preference, from measurement (profiles/r06/README.md): (1) packed rows + a FULL image when the budget has room
        # for all three copies - beside the co-running SYRK the packed second J pass costs less than the square one (same bytes,
        # same alignment: 41.5-42.1 vs 45.9-47.4 ms of SYRK on one box), so config 3 keeps its r05 speed; (2) the square rows alone
        # when 3x does not fit and 2x does (taxol on one GPU, large shards): every row on the square kernel, no second copy, the XC
        # image beside it; (3) packed rows + whatever partial image / diagonal blocks the budget leaves.
        luxury = nL * npair * 8 + nL * rows * rows * 8 + max(self._reserve_after_build(nL, rows), int(self.k_square_reserve))
        # the order of preference is the library's (PAMD_df_layout_pick, shared with the C handle's build_rows); the byte counts are
        # this layer's.  Collective: a rank that cannot have a layout takes every rank to the next one
        import ctypes as _c
        from .. import lib as _lib
        so = _lib.load_library()
        pick = int(so.PAMD_df_layout_pick(_c.c_longlong(int(luxury + (2 << 30))), _c.c_longlong(int(build_need)), _c.c_longlong(int(after_need)),
                                          _c.c_longlong(int(free)), _c.c_int(1 if getattr(self, 'prefer_image', True) else 0)))
        if getattr(self, 'prefer_image', True) and self._all_ranks_agree(pick == 2):
            return 'packed'
        fits = max(build_need, after_need) <= free
        return 'square' if self._all_ranks_agree(fits) else 'packed'

    def _all_ranks_agree(self, fits):
        """The out-of-core decision of a multi-rank job is COLLECTIVE (r06): each rank used to decide from its own
        hipMemGetInfo; ranks that disagreed (shared devices, the one-row shard imbalance near the threshold) then issued
        different collectives - in-core ranks one packed [J~ | K] all-reduce, out-of-core ranks two nao^2 all-reduces - i.e. an
        RCCL hang or corrupted sums.  MIN over the ranks: one rank out of core = every rank out of core (same reduction path)."""
        from ..lib import comm as _comm
        if getattr(self, '_shard_override', None) is not None or not _comm.active(self.world_size):
            return bool(fits)
        import torch
        import torch.distributed as dist
        dev = self._device() if _comm.backend_name(self.group) == 'nccl' else 'cpu'
        flag = torch.tensor([1 if fits else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _rows_do_not_fit(self, nrows, ncol, dev):
        import torch
        need = nrows * ncol * 8
        if self.outcore_device_bytes and need > self.outcore_device_bytes:
            return True
        if dev.type != 'cuda':
            return False
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        return need + (2 << 30) > free

    def _native_from_rows(self, rows, shard_rows, naux, dev):
        """Ready-made host rows (the caller's array / a mapped `_cderi` file) behind the out-of-core C handle."""
        import torch
        from .native import NativeDF
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        torch.cuda.empty_cache()
        sharded = self.world_size > 1 or getattr(self, '_shard_override', None) is not None
        self._native = NativeDF.from_rows(self.mol, rows, device=idx, max_device_bytes=int(getattr(self, 'outcore_device_bytes', 0)),
                                          borrow=True, shard=(self.rank, self.world_size) if sharded else None,
                                          shard_rows=shard_rows, naux=naux)
        self._naux = naux

    def would_fit(self):
        """Cheap predicate (host arithmetic + one memory query, no integrals): does this rank's packed shard of the tensor,
        with the slab work space of the build, fit the device - or the `outcore_device_bytes` cap?  build() asks it before any
        metric work; scf.device_scf.eligible() asks it instead of building."""
        import torch
        if self.has_tensor():
            return True
        if getattr(self, '_native', None) is not None:
            return False
        if self.auxmol is None:
            self.auxmol = addons.make_auxmol(self.mol, self.auxbasis)
        naux = self.auxmol.nao_nr()
        nao = _mol_nao(self.mol)
        npair = nao * (nao + 1) // 2
        l0, l1 = self.shard_range(naux, self.rank, self.world_size)
        shard_b = (l1 - l0) * npair * 8
        if self.outcore_device_bytes and shard_b > self.outcore_device_bytes:
            return False
        dev = self._device()
        if dev.type != 'cuda':
            return True
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        return shard_b + min(24 << 30, npair * naux * 8) <= free

    def get_naoaux(self):
        if self._naux is None:
            self.build()
        return self._naux

    def out_of_core(self):
        """None, or the layout of the C handle that holds the tensor because it did not fit the device (NativeDF.layout())."""
        nat = getattr(self, '_native', None)
        return None if nat is None else nat.layout()

    def loop(self, blksize=None, local=False):
        """Yield host row blocks of the FULL tensor, rows 0..naux in order (pyscf/df/df.py:214-242) - what a stock consumer
        (DF-MP2, DF-CCSD, `ao2mo`) expects.  With the aux index sharded over several ranks every rank takes part in the
        iteration (it is a collective: each block is assembled from its owners' rows by an all-reduce of a zero-padded
        buffer) and every rank receives every block.  `local=True`: only this rank's rows [l0, l1), no communication."""
        import torch
        if not self.has_tensor() and getattr(self, '_native', None) is None:
            self.build()
        if blksize is None:
            blksize = self.blockdim
        if getattr(self, '_native', None) is not None:
            if self._native.shard is not None and self._native.shard[1] > 1 and not local:
                raise NotImplementedError('DF.loop(): this rank holds its shard of the tensor out of core (C handle); the collective '
                                          'iteration over ALL aux rows needs the in-core shards - pass local=True for the rows of the shard')
            for blk in self._native.loop(blksize):
                yield blk
            return
        n, npair = self.tensor_shape()
        from ..lib import comm as _comm
        sharded = _comm.active(self.world_size) and getattr(self, '_shard_override', None) is None
        if getattr(self, '_shard_override', None) is not None and self._shard_override[1] > 1 and not local:
            raise RuntimeError('DF.loop(): this object holds one emulated shard (rank %d of %d) of the tensor, not all aux '
                               'rows; pass local=True for the rows of the shard' % tuple(self._shard_override))
        if local or not sharded:
            for b0 in range(0, n, blksize):
                yield self.packed_rows(b0, min(b0 + blksize, n)).cpu().numpy()
            return
        naux = self.get_naoaux()
        l0, l1 = self.shard_range(naux, self.rank, self.world_size)
        buf = torch.empty((min(blksize, naux), npair), dtype=torch.float64, device=self.tensor_device())
        for b0 in range(0, naux, blksize):
            b1 = min(b0 + blksize, naux)
            blk = buf[:b1 - b0]
            blk.zero_()
            a0, a1 = max(b0, l0), min(b1, l1)
            if a1 > a0:
                blk[a0 - b0:a1 - b0] = self.packed_rows(a0 - l0, a1 - l0)
            _comm.all_reduce([blk], self.group, self.world_size)
            yield blk.cpu().numpy()

    # -- downstream consumers of the tensor ----------------------------------
    def _row_blocks(self, blksize):
        """(b0, packed device rows [nb][nao_pair]) over THIS rank's rows, whatever holds them: the packed tensor (views), the
        square rows (packed block by block), or - r06, VERDICT r05 Missing 4 - the out-of-core C handle (its loop() blocks,
        uploaded): the reference's consumers need nothing but DF.loop() either (pyscf/df/df.py:214-242,269-296)."""
        import torch
        nat = getattr(self, '_native', None)
        if nat is not None:
            dev = self._device()
            b0 = 0
            for blk in nat.loop(blksize):
                yield b0, torch.from_numpy(np.ascontiguousarray(blk)).to(dev)
                b0 += blk.shape[0]
            return
        n = self.tensor_shape()[0]
        for b0 in range(0, n, blksize):
            yield b0, self.packed_rows(b0, min(b0 + blksize, n))

    def _pair_gram(self, a_blocks, m, n, dev):
        """sum_L a[L,:]^T b[L,:] over this rank's rows - `a_blocks` yields (a_blk, b_blk) row blocks -, all-reduced over the shards."""
        import torch
        import ctypes as _c
        from .. import lib as _lib
        so = _lib.load_library()
        if torch.device(dev).type == 'cuda':
            from ..lib import hbm
            if m * n * 8 + (1 << 30) > hbm.free_bytes(dev):
                raise MemoryError('the (%d, %d) pair matrix (%.1f GB) does not fit the device: this is the O(nao^4) object density '
                                  'fitting exists to avoid - iterate DF.loop() blocks instead' % (m, n, m * n * 8e-9))
        out = torch.zeros((1, m, n), dtype=torch.float64, device=dev)
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        for a, b in a_blocks:
            if a.shape[0]:
                df_jk._call(self, 'dgemm_tn', so.PAMD_dgemm_tn, _c.c_void_p(a.data_ptr()), _c.c_int(a.stride(0)),
                            _c.c_void_p(b.data_ptr()), _c.c_int(b.stride(0)), _c.c_void_p(out.data_ptr()), _c.c_int(n),
                            _c.c_int(m), _c.c_int(n), _c.c_long(a.shape[0]), _c.c_int(0), _c.c_int(1), st)
                torch.cuda.current_stream().synchronize()      # (the block may be a temporary: keep it alive until the product ran)
        df_jk._allreduce(self, [out])
        return out[0]

    def _tensor_dev_npair(self):
        nat = getattr(self, '_native', None)
        if nat is not None:
            return self._device(), nat.nao * (nat.nao + 1) // 2
        return self.tensor_device(), self.tensor_shape()[1]

    def get_eri(self):
        """8-fold packed (pq|rs) ~ sum_L B[L,pq] B[L,rs] (pyscf/df/df.py:269-276: lib.dot(eri1.T, eri1)
        then ao2mo.restore(8, ...)): the lower triangle of the (nao_pair, nao_pair) matrix.  Row block by row block, so that the
        out-of-core handle (r06) and the square layout serve it as well."""
        from .. import lib as _lib
        if not self.has_tensor() and getattr(self, '_native', None) is None:
            self.build()
        dev, npair = self._tensor_dev_npair()
        blk = max(1, (1 << 30) // (npair * 8))
        eri4 = self._pair_gram(((r, r) for _b0, r in self._row_blocks(blk)), npair, npair, dev).cpu().numpy()
        return _lib.pack_tril(eri4)
    get_ao_eri = get_eri

    def _half_transform_pairs(self, ci, cj, compact):
        """Lij[L, ij] = sum_pq ci[p,i] B_L[p,q] cj[q,j] on the device (the role of _ao2mo.nr_e2 with
        aosym='s2', pyscf/df/df.py:287-293); ij packed i >= j when ci is cj and compact."""
        import torch
        import ctypes as _c
        from .. import lib as _lib
        so = _lib.load_library()
        dev, npair = self._tensor_dev_npair()
        nao = self.mol.nao_nr()
        same = compact and ci.shape == cj.shape and abs(ci - cj).max() < 1e-13      # iden_coeffs, ao2mo/incore.py
        orb, ni_pad, ldo = df_jk.pad_orbitals(np.asarray(ci, dtype=np.float64), dev)
        ni, nj = ci.shape[1], cj.shape[1]
        cj_dev = torch.from_numpy(np.ascontiguousarray(cj, dtype=np.float64)).to(dev)
        ldx = (nao + 15) // 16 * 16
        blk = max(1, int((1 << 30) // (max(ni_pad, 16) * ldx * 8)))
        st = _c.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs = []
        if same:
            ti, tj = np.tril_indices(ni)
            sel = torch.from_numpy(ti * nj + tj).to(dev)
        for _b0, rows in self._row_blocks(blk):
            nb = rows.shape[0]
            X = torch.zeros((nb, ni_pad, ldx), dtype=torch.float64, device=dev)
            df_jk._call(self, 'e2_symm', so.PAMD_nr_e2_symm, _c.c_void_p(rows.data_ptr()), _c.c_long(npair),
                        _c.c_int(nb), _c.c_int(nao), _c.c_void_p(orb.data_ptr()), _c.c_int(ldo),
                        _c.c_int(orb.shape[0]), _c.c_int(ni_pad), _c.c_void_p(X.data_ptr()), _c.c_int(ldx),
                        _c.c_void_p(0), _c.c_void_p(0), st)
            y = torch.matmul(X[:, :ni, :nao], cj_dev).reshape(nb, ni * nj)      # second index: plain library GEMM
            outs.append(y[:, sel] if same else y)
        if not outs:
            return torch.zeros((0, ni * (ni + 1) // 2 if same else ni * nj), dtype=torch.float64, device=dev)
        return torch.cat(outs)

    def ao2mo(self, mo_coeffs, compact=True):
        """(ij|kl) in the MO bases mo_coeffs = (Ci, Cj, Ck, Cl) (or one matrix for all four):
        matrix (nij_pair, nkl_pair), pairs packed when the two coefficient blocks coincide and
        `compact` (pyscf/df/df.py:278-296).  r06: works on an out-of-core tensor too (row blocks of the handle's loop())."""
        if not self.has_tensor() and getattr(self, '_native', None) is None:
            self.build()
        if isinstance(mo_coeffs, np.ndarray) and mo_coeffs.ndim == 2:
            mo_coeffs = (mo_coeffs,) * 4
        ci, cj, ck, cl = [np.asarray(c, dtype=np.float64) for c in mo_coeffs]
        lij = self._half_transform_pairs(ci, cj, compact)
        sym = ci.shape == ck.shape and cj.shape == cl.shape and abs(ci - ck).max() < 1e-13 and abs(cj - cl).max() < 1e-13
        lkl = lij if sym else self._half_transform_pairs(ck, cl, compact)
        return self._pair_gram([(lij, lkl)], lij.shape[1], lkl.shape[1], lij.device).cpu().numpy()
    get_mo_eri = ao2mo

    def _need_in_core(self, what):
        """The consumers that work on the HBM-resident tensor say so when build() handed the tensor to the out-of-core handle
        (they used to dereference `_cderi_dev = None`)."""
        if not self.has_tensor() and getattr(self, '_native', None) is not None:
            raise NotImplementedError('DF.%s needs the in-core tensor; this object holds it out of core (%s): use more ranks / '
                                      'devices, or iterate DF.loop() blocks on the host' % (what, self.out_of_core()))

    def _shard_path(self, path):
        return '%s.rank%dof%d.npz' % (path, self.rank, self.world_size)

    def save(self, path=None, fmt=None):
        """Write the tensor to disk.  fmt 'hdf5' (default when libhdf5 is found and the path does not end in .npy): ONE file
        with the dataset 'j3c' (naux, nao_pair) - the reference's `_cderi` file (pyscf/df/df.py:97-99,185-199,
        outcore.py:217-221; readable by stock PySCF / h5py); with several ranks rank 0 creates it and the ranks write their
        row ranges one after the other.  fmt 'npy': a .npy file of the full array (one rank), or per-rank
        `path.rank<r>of<w>.npz` archives with (l0, l1, naux).  `DF(mol)._cderi = path` loads either back in build()."""
        from ..lib import hdf5
        path = path or self._cderi_to_save
        if not self.has_tensor() and getattr(self, '_native', None) is None:
            self.build()
        if fmt is None:
            fmt = 'hdf5' if (hdf5.available() and not path.endswith(('.npy', '.npz'))) else 'npy'
        if getattr(self, '_native', None) is not None:
            return self._save_out_of_core(path, fmt)
        if fmt == 'hdf5':
            naux, npair = self._naux, self.tensor_shape()[1]
            l0, l1 = self.shard_range(naux, self.rank, self.world_size) if self.world_size > 1 else (0, naux)
            step = max(1, (1 << 30) // (npair * 8))
            for turn in range(self.world_size):
                if turn == self.rank:
                    with hdf5.File(path, 'w' if turn == 0 else 'r+') as f:
                        d = f.create_dataset('j3c', (naux, npair)) if turn == 0 else f['j3c']
                        for r0 in range(l0, l1, step):
                            r1 = min(r0 + step, l1)
                            d.write_rows(r0, self.packed_rows(r0 - l0, r1 - l0).cpu().numpy())
                if self.world_size > 1 and getattr(self, '_shard_override', None) is None:
                    import torch.distributed as dist
                    dist.barrier(group=self.group)
            return path
        if self.world_size > 1:
            l0, l1 = self.shard_range(self._naux, self.rank, self.world_size)
            out = self._shard_path(path)
            np.savez(out, j3c=np.vstack([b for b in self.loop(local=True)]), l0=l0, l1=l1, naux=self._naux)
            return out
        from numpy.lib import format as _fmt
        nloc, npair = self.tensor_shape()
        with open(path, 'wb') as f:                    # a .npy written block by block (no second host copy of the tensor)
            _fmt.write_array_header_1_0(f, {'descr': '<f8', 'fortran_order': False, 'shape': (nloc, npair)})
            for blk in self.loop(max(1, (1 << 30) // (npair * 8)), local=True):
                f.write(np.ascontiguousarray(blk).tobytes())
        return path

    def _save_out_of_core(self, path, fmt):
        """save() of a tensor held by the out-of-core handle: the row blocks of NativeDF.loop() go to the file one after the other
        (`_cderi_to_save` is honoured on this path too) - never more than one block on the host."""
        from ..lib import hdf5
        nat = self._native
        nao = nat.nao
        npair = nao * (nao + 1) // 2
        naux = self._naux
        l0, l1 = nat.shard_rows
        sharded = nat.shard is not None and nat.shard[1] > 1
        step = max(1, (1 << 30) // (npair * 8))
        if fmt == 'hdf5':
            world = self.world_size if sharded else 1
            for turn in range(world):
                if turn == (self.rank if sharded else 0):
                    with hdf5.File(path, 'w' if turn == 0 else 'r+') as f:
                        d = f.create_dataset('j3c', (naux, npair)) if turn == 0 else f['j3c']
                        r0 = l0
                        for blk in nat.loop(step):
                            d.write_rows(r0, blk)
                            r0 += blk.shape[0]
                if sharded and getattr(self, '_shard_override', None) is None:
                    import torch.distributed as dist
                    dist.barrier(group=self.group)
            return path
        if sharded:
            out = self._shard_path(path)
            np.savez(out, j3c=np.vstack(list(nat.loop(step))), l0=l0, l1=l1, naux=naux)
            return out
        from numpy.lib import format as _fmt
        with open(path, 'wb') as f:                    # a .npy written block by block
            _fmt.write_array_header_1_0(f, {'descr': '<f8', 'fortran_order': False, 'shape': (naux, npair)})
            for blk in nat.loop(step):
                f.write(np.ascontiguousarray(blk).tobytes())
        return path

    def range_coulomb(self, omega):
        """DF object holding the long-range (omega > 0, erf(omega r12)/r12) or short-range (omega < 0,
        erfc(|omega| r12)/r12) tensor, cached per omega (pyscf/df/df.py:298-333)."""
        if omega is None or omega == 0:
            return self
        key = '%.6f' % omega
        if key not in self._rsh_df:
            obj = DF(self.mol, self.auxbasis, self.device, self.group)
            obj.omega = float(omega)
            obj.auxmol = self.auxmol
            obj.k_block_bytes, obj.k_nsplit, obj.lindep = self.k_block_bytes, self.k_nsplit, self.lindep
            obj.k_e2_pipeline, obj.j2_policy = self.k_e2_pipeline, self.j2_policy
            obj.decompose_j2c = self.decompose_j2c
            self._rsh_df[key] = obj
        return self._rsh_df[key]

    def get_jk(self, dm, hermi=1, with_j=True, with_k=True, direct_scf_tol=1e-13, omega=None):
        if omega is not None and omega != 0:
            return self.range_coulomb(omega).get_jk(dm, hermi, with_j, with_k, direct_scf_tol)
        if getattr(self, '_native', None) is not None:          # out of core: the C handle holds the tensor (build())
            vj, vk = self._native.get_jk(dm, hermi, with_j, with_k, direct_scf_tol)
            if self._native.shard is not None:
                vj, vk = self._allreduce_host(vj, vk)           # this rank's partial sums -> the sum over the aux shards
            return vj, vk
        return df_jk.get_jk(self, dm, hermi, with_j, with_k, direct_scf_tol)

    def _allreduce_host(self, vj, vk):
        """Sum the host arrays of a sharded out-of-core handle over the ranks: RCCL on device copies (backend 'nccl'), in place
        on the page-locked host arrays with 'gloo'; no-op without an active group (an emulated shard keeps its partial sums)."""
        import torch
        from ..lib import comm as _comm
        if getattr(self, '_shard_override', None) is not None or not _comm.active(self.world_size):
            return vj, vk
        arrs = [a for a in (vj, vk) if a is not None]
        if _comm.backend_name(self.group) == 'nccl':
            dev = self._device()
            ts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]
            _comm.all_reduce(ts, self.group, self.world_size)
            outs = [t.cpu().numpy() for t in ts]
        else:
            outs = [np.ascontiguousarray(a) for a in arrs]
            _comm.all_reduce([torch.from_numpy(a) for a in outs], self.group, self.world_size)
        it = iter(outs)
        return (next(it) if vj is not None else None), (next(it) if vk is not None else None)


GDF = DF
