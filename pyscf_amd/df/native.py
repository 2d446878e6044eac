"""numpy-only client of the host-array C ABI (include/pyscf_amd.h: PAMD_df_create / PAMD_df_get_jk / PAMD_df_export_cderi).

This is the binding a PySCF maintainer would write: ctypes on raw numpy buffers, the caller owns every array, no torch, no
device pointers in Python - the same convention as the reference's own C calls (pyscf/df/df_jk.py:373-379).  ``NativeDF``
duck-types the part of ``pyscf.df.DF`` that ``_DFHF.get_jk`` uses (``get_jk``, ``get_naoaux``, ``loop``, ``build``, ``reset``,
pyscf/df/df.py:147-267), so ``mf.with_df = NativeDF(mol)`` works with any object that has libcint-format ``_atm/_bas/_env``.

(The torch-resident ``pyscf_amd.df.DF`` stays the production object: it shards over ranks, keeps results on the device for
the HBM-resident SCF loop, and shares the stream with the XC path.  This one exists for callers without a device runtime.)
"""
import ctypes
import os

import numpy as np

_c = ctypes
_lib = None


def load():
    """libpyscf_amd.so loaded WITHOUT importing torch (the HIP runtime is then the system one the library links by SONAME)."""
    global _lib
    if _lib is None:
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        so = os.environ.get('PAMD_LIBRARY') or os.path.join(here, 'lib', 'libpyscf_amd.so')
        if not os.path.exists(so):
            raise ImportError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"`' % so)
        _preload_hip_runtime()
        lib = _c.CDLL(so)
        lib.PAMD_last_error.restype = _c.c_char_p
        _lib = lib
    return _lib


def _preload_hip_runtime():
    """ONE HIP runtime per process.  libpyscf_amd.so names libamdhip64.so.7 by SONAME only; a PyTorch-ROCm wheel bundles its own copy
    under torch/lib.  If this library pulled in the system runtime first and the process imported torch afterwards (a script that
    builds a NativeDF and then uses pyscf_amd.scf), the second runtime found no device: "the integral engine needs a HIP device"
    after a 560 GB build (r04's lost config-5 SCF, misread then as a memory problem).  So when a torch install exists but is not
    imported yet, its bundled runtime is loaded here - WITHOUT importing torch - and a later `import torch` shares it; without a
    torch install the system runtime is used.  PAMD_HIP_RUNTIME=<path> | system overrides."""
    import sys
    choice = os.environ.get('PAMD_HIP_RUNTIME', '')
    if choice == 'system' or 'torch' in sys.modules:
        return
    path = choice
    if not path:
        try:
            import importlib.util
            spec = importlib.util.find_spec('torch')
            if spec is not None and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
                path = cand if os.path.exists(cand) else ''
        except Exception:
            path = ''
    if path:
        try:
            _c.CDLL(path, mode=_c.RTLD_GLOBAL)
        except OSError:
            pass


def _check(rc):
    if rc != 0:
        raise RuntimeError('libpyscf_amd call failed (%d): %s' % (rc, load().PAMD_last_error().decode()))


class _PinnedBlock:
    """One PAMD_host_alloc block (page-locked, portable); freed with the last numpy view of it."""

    def __init__(self, nbytes):
        self.ptr = _c.c_void_p()
        _check(load().PAMD_host_alloc(_c.c_longlong(int(nbytes)), _c.byref(self.ptr)))
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                load().PAMD_host_free(self.ptr)
        except Exception:
            pass


def _alloc_pinned(nbytes):
    blk = _PinnedBlock(nbytes)
    return blk.ptr.value, blk


_pool = None


def pinned_empty(shape):
    """float64 array of `shape` in page-locked host memory (results of get_jk / nr_rks: the device -> host copy then runs at the
    PCIe rate, ~50 GB/s, instead of the pageable ~10 GB/s plus first-touch page faults).  Blocks are recycled through an explicit
    free list (pyscf_amd/lib/pinned.py): one is reused only when the array handed out from it AND every view of that array are
    gone, so callers keep and modify results as long as they like.  Falls back to pageable numpy memory if the allocation is
    refused."""
    global _pool
    if _pool is None:
        from ..lib.pinned import PinnedPool
        _pool = PinnedPool(_alloc_pinned)
    n = int(np.prod(shape)) if len(shape) else 1
    got = _pool.take(n)
    if got is None:
        return np.empty(shape)
    return got[0][:n].reshape(shape)


def _conc_env(atm1, bas1, env1, atm2, bas2, env2):
    """gto.conc_env (pyscf/gto/mole.py:805-838): one atm / bas / env holding both molecules, pointers of the second shifted."""
    off = len(env1)
    natm_off = len(atm1)
    atm2 = np.array(atm2, dtype=np.int32, copy=True)
    bas2 = np.array(bas2, dtype=np.int32, copy=True)
    atm2[:, 1] += off          # PTR_COORD
    atm2[:, 3] += off          # PTR_ZETA
    bas2[:, 0] += natm_off     # ATOM_OF
    bas2[:, 5] += off          # PTR_EXP
    bas2[:, 6] += off          # PTR_COEFF
    return (np.ascontiguousarray(np.vstack((atm1, atm2)), dtype=np.int32), np.ascontiguousarray(np.vstack((bas1, bas2)), dtype=np.int32),
            np.ascontiguousarray(np.hstack((env1, env2)), dtype=np.float64))


def _scaled_occupied(c, occ):
    """C[:, occ > 0] * sqrt(occ) as a C-contiguous (nao, nocc) array; a contiguous occupied range (the usual case) is a slice, not
    a boolean-mask gather over the whole coefficient matrix (2-4 ms at nao = 1856)."""
    idx = np.flatnonzero(occ > 0)
    if len(idx) and idx[-1] - idx[0] + 1 == len(idx):
        return np.ascontiguousarray(c[:, idx[0]:idx[-1] + 1] * np.sqrt(occ[idx[0]:idx[-1] + 1]))
    return np.ascontiguousarray(c[:, idx] * np.sqrt(occ[idx]))


class _Options(_c.Structure):
    """PAMD_df_options of include/pyscf_amd.h"""
    _fields_ = [('lindep', _c.c_double), ('omega', _c.c_double), ('devices', _c.POINTER(_c.c_int)), ('ndev', _c.c_int),
                ('flags', _c.c_int), ('max_device_bytes', _c.c_longlong), ('part', _c.c_int), ('nparts', _c.c_int),
                ('reserve_bytes', _c.c_longlong)]


class NativeDF:
    """``NativeDF(mol, auxbasis)``: one GPU.  ``NativeDF(mol, auxbasis, devices=range(8))``: the same object over several
    devices of the node in THIS process - the aux index is sharded inside the C handle (PAMD_df_create_multi), a stock
    single-process ``mf.with_df = NativeDF(mol, devices=range(8)); mf.kernel()`` uses all of them.  ``max_device_bytes``
    caps the HBM a part may take: rows beyond it (or beyond the device's free memory) are kept in page-locked host memory
    and streamed under the kernels in every build (out-of-core, PCIe-bound for those rows)."""
    blockdim = 240

    def __init__(self, mol, auxbasis=None, auxmol=None, device=0, lindep=1e-7, devices=None, omega=0.0, max_device_bytes=0,
                 shard=None):
        self.mol = mol
        self.auxbasis = auxbasis
        self.auxmol = auxmol
        self.device = device
        self.devices = None if devices is None else [int(d) for d in devices]
        self.lindep = lindep
        self.omega = float(omega)
        self.max_device_bytes = int(max_device_bytes)
        # (rank, world): this object holds ONE RANK's rows of the aux-sharded tensor (a multi-process job, one process per GPU);
        # get_jk then returns that shard's PARTIAL J/K and the caller sums over the ranks (pyscf_amd.df.DF does, RCCL all-reduce)
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        self._h = None
        self._naux = None
        self._rsh_df = {}                         # omega -> NativeDF of that operator (pyscf/df/df.py:298-333 range_coulomb)
        # r06 - one HBM budget: bytes per device the XC leg of the same calculation will cache (Kohn-Sham objects set it in
        # density_fit); the handle holds the rows in the square layout (2x, no second copy) only when that still fits
        self.xc_image_hint = 0

    def build(self):
        if self._h is not None:
            return self
        if self.auxmol is False:
            raise RuntimeError('this NativeDF was made from ready-made rows (from_rows) and has been reset: make a new one')
        if self.auxmol is None:
            from . import addons                  # host-only: basis tables (any object with _atm/_bas/_env works as auxmol)
            self.auxmol = addons.make_auxmol(self.mol, self.auxbasis)
        mol, aux = self.mol, self.auxmol
        atm, bas, env = _conc_env(np.asarray(mol._atm), np.asarray(mol._bas), np.asarray(mol._env),
                                  np.asarray(aux._atm), np.asarray(aux._bas), np.asarray(aux._env))
        h = _c.c_void_p()
        devs = self.devices if self.devices is not None else [int(self.device)]
        arr = (_c.c_int * len(devs))(*devs)
        reserve = 0
        if self.xc_image_hint:
            # the compact image is dealt over the parts; its work buffers (12 GB) exist on every device
            reserve = int(self.xc_image_hint) // max(len(set(devs)), 1) + (12 << 30)
        flags = (1 if self.devices is not None else 0) | (2 if self.shard is not None else 0) | (4 if reserve else 0)
        part, nparts = self.shard if self.shard is not None else (0, 1)
        opt = _Options(self.lindep, self.omega, arr, len(devs), flags, self.max_device_bytes, part, nparts, reserve)
        _check(load().PAMD_df_create_ex(atm.ctypes.data_as(_c.c_void_p), _c.c_int(len(atm)), bas.ctypes.data_as(_c.c_void_p),
                                        _c.c_int(len(mol._bas)), _c.c_int(len(aux._bas)), env.ctypes.data_as(_c.c_void_p),
                                        _c.c_int(len(env)), _c.byref(opt), _c.byref(h)))
        self._h = h
        n = _c.c_int()
        info = (_c.c_int * 4)()
        _check(load().PAMD_df_shard_info(h, info))
        self.shard_rows = (info[0], info[0] + info[1])      # global rows this handle holds ([0, naux) unless `shard` is set)
        self._naux = info[2]
        _check(load().PAMD_df_nao(h, _c.byref(n)))
        self.nao = n.value
        return self
    kernel = build

    @classmethod
    def from_rows(cls, mol, rows, device=0, max_device_bytes=0, borrow=True, shard=None, shard_rows=None, naux=None):
        """A handle over tensor rows the caller already holds (PAMD_df_create_from_rows): `rows` (nrows, nao_pair) float64 C-order in
        host memory - a numpy array or an np.memmap of the 'j3c' dataset of a PySCF `_cderi` file.  What fits the device is
        uploaded, the rest is streamed under the kernels in every build - with `borrow` straight out of `rows` (kept alive by
        this object; nothing the size of the tensor is allocated on the host).  `shard` / `shard_rows` / `naux`: the rows are
        one rank's shard [r0, r1) of a tensor of `naux` rows (get_jk then returns partial sums)."""
        rows = np.asarray(rows) if not isinstance(rows, np.memmap) else rows
        if rows.dtype != np.float64 or rows.ndim != 2 or not rows.flags.c_contiguous:
            raise ValueError('from_rows: a C-contiguous float64 (nrows, nao_pair) array')
        from .df import _mol_nao                       # (host-only helper; importing the module pulls in no torch)
        nao = _mol_nao(mol)
        if rows.shape[1] != nao * (nao + 1) // 2:
            raise ValueError('from_rows: %d columns, expected nao_pair = %d' % (rows.shape[1], nao * (nao + 1) // 2))
        self = cls(mol, device=device, max_device_bytes=max_device_bytes, shard=shard)
        h = _c.c_void_p()
        _check(load().PAMD_df_create_from_rows(_c.c_void_p(rows.ctypes.data), _c.c_int(rows.shape[0]), _c.c_int(nao), _c.c_int(int(device)),
                                               _c.c_longlong(int(max_device_bytes)), _c.c_int(1 if borrow else 0), _c.byref(h)))
        self._h = h
        self._rows_keepalive = rows if borrow else None
        self.nao = nao
        self.shard_rows = tuple(shard_rows) if shard_rows is not None else (0, rows.shape[0])
        self._naux = int(naux) if naux is not None else rows.shape[0]
        self.auxmol = False                        # (no auxiliary molecule: the rows came ready-made)
        return self

    def last_timing(self):
        """Host-clock timings of the last get_jk inside the handle (PAMD_df_last_timing): {'parts', 'sum_download_ms', 'peer',
        'compute_ms': [...], 'push_ms': [...], 'push_bytes': [...], 'e2_ms': [...], 'syrk_ms': [...]} - what bench.py
        --single-process reports as `roofline` and `comm`."""
        out = (_c.c_double * (3 + 5 * 64))()
        _check(load().PAMD_df_last_timing(self._h, out, _c.c_int(len(out))))
        n = int(out[0])
        return dict(parts=n, sum_download_ms=out[1], peer=int(out[2]), compute_ms=[out[3 + 5 * i] for i in range(n)],
                    push_ms=[out[4 + 5 * i] for i in range(n)], push_bytes=[int(out[5 + 5 * i]) for i in range(n)],
                    e2_ms=[out[6 + 5 * i] for i in range(n)], syrk_ms=[out[7 + 5 * i] for i in range(n)])

    def layout(self):
        """{'parts', 'rows_resident', 'rows_host', 'rows_square', 'peer', 'part_rows'} of the built handle (PAMD_df_layout)."""
        self.build()
        lay = (_c.c_long * 5)()
        rows = (_c.c_int * 64)()
        _check(load().PAMD_df_layout(self._h, lay, rows))
        return dict(parts=lay[0], rows_resident=lay[1], rows_host=lay[2], rows_square=lay[3], peer=lay[4],
                    part_rows=[rows[i] for i in range(lay[0])],
                    tensor_layout='square' if load().PAMD_df_tensor_layout(self._h) else 'packed')

    def range_coulomb(self, omega):
        """The handle of erf(omega r12)/r12 (omega > 0) or erfc(|omega| r12)/r12 (omega < 0), cached per omega."""
        if not omega:
            return self
        if self.auxmol is False:
            #  ready-made rows carry no auxiliary basis: the attenuated tensor cannot be derived from them
            raise NotImplementedError('NativeDF.range_coulomb(): this handle was made from ready-made tensor rows (from_rows / a '
                                      '_cderi file); a range-separated tensor needs the integrals - build a NativeDF(mol, auxbasis)')
        key = '%.6f' % omega
        if key not in self._rsh_df:
            self._rsh_df[key] = NativeDF(self.mol, self.auxbasis, self.auxmol, self.device, self.lindep, self.devices, omega,
                                         self.max_device_bytes, self.shard)
        return self._rsh_df[key]

    def reset(self, mol=None):
        if mol is not None and self.auxmol is False:
            #  reset(mol) used to set auxmol = None: a later build() then silently recomputed the tensor from integrals
            # instead of the caller's rows
            raise NotImplementedError('NativeDF.reset(mol): this handle was made from ready-made tensor rows (from_rows); make a '
                                      'new NativeDF for the new molecule')
        if self._h is not None:
            load().PAMD_df_destroy(self._h)
        self._h = None
        self._naux = None
        self._rows_keepalive = None
        for o in getattr(self, '_rsh_df', {}).values():
            o.reset()
        self._rsh_df = {}
        if mol is not None:
            self.mol = mol
            self.auxmol = None
        return self

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    def get_naoaux(self):
        self.build()
        return self._naux

    def loop(self, blksize=None):
        self.build()
        blksize = blksize or self.blockdim
        npair = self.nao * (self.nao + 1) // 2
        nrow = self.shard_rows[1] - self.shard_rows[0]       # (a rank's shard: its own rows, local numbering)
        for b0 in range(0, nrow, blksize):
            b1 = min(b0 + blksize, nrow)
            out = np.empty((b1 - b0, npair))
            _check(load().PAMD_df_export_cderi(self._h, _c.c_int(b0), _c.c_int(b1), out.ctypes.data_as(_c.c_void_p)))
            yield out

    def device_index(self):
        """HIP device of a one-part handle (None for a device list of several parts)."""
        devs = self.devices if self.devices is not None else [int(self.device)]
        return int(devs[0]) if len(devs) == 1 else None

    def get_jk_device(self, dm_dev, orbo_dev, with_k=True):
        """J and K of ONE closed-shell density with inputs and outputs in HBM (r06, VERDICT r05 item 6: PAMD_df_get_jk with flags
        bit 3 - device pointers): dm_dev (nao, nao) and orbo_dev = C_occ sqrt(occ) (nao, nocc), contiguous float64 torch tensors on
        the handle's device with dm_dev = orbo_dev orbo_dev^T (the caller built it so: flags bit 0); returns (vj, vk | None) as
        torch tensors.  What scf/device_scf.py calls when `with_df` is this handle: the HBM-resident SCF loop over a handle-held
        tensor."""
        import torch
        self.build()
        nao = dm_dev.shape[-1]
        dm_dev = dm_dev.contiguous()
        vj = torch.empty((nao, nao), dtype=torch.float64, device=dm_dev.device)
        vk = torch.empty((nao, nao), dtype=torch.float64, device=dm_dev.device) if with_k else None
        orbo = nocc = None
        if with_k:
            orbo = orbo_dev.contiguous()
            nocc = np.array([orbo.shape[1]], dtype=np.int32)
        # the handle works on its own streams: the inputs must be complete before it reads them, and it returns when J and K are
        torch.cuda.current_stream(dm_dev.device).synchronize()
        _check(load().PAMD_df_get_jk(
            self._h, _c.c_void_p(dm_dev.data_ptr()), _c.c_void_p(orbo.data_ptr()) if with_k else None,
            nocc.ctypes.data_as(_c.c_void_p) if with_k else None, _c.c_int(1), _c.c_int(nao), _c.c_int(1), _c.c_int(1),
            _c.c_int(int(with_k)), _c.c_int((1 if with_k else 0) | 8), _c.c_void_p(vj.data_ptr()),
            _c.c_void_p(vk.data_ptr()) if with_k else None))
        return vj, vk

    def get_jk(self, dm, hermi=1, with_j=True, with_k=True, direct_scf_tol=1e-13, omega=None):
        if omega is not None and omega != 0 and omega != self.omega:
            return self.range_coulomb(omega).get_jk(dm, hermi, with_j, with_k, direct_scf_tol)
        self.build()
        dms = np.asarray(dm)
        if np.iscomplexobj(dms):
            vjr, vkr = self.get_jk(dms.real, 0, with_j, with_k)
            vji, vki = self.get_jk(dms.imag, 0, with_j, with_k)
            return (vjr + 1j * vji if with_j else None), (vkr + 1j * vki if with_k else None)
        shape = dms.shape
        nao = shape[-1]
        dms = np.ascontiguousarray(dms.reshape(-1, nao, nao), dtype=np.float64)
        nset = len(dms)
        orbo = nocc = None
        flags = 0
        mo_coeff = getattr(dm, 'mo_coeff', None)
        if with_k and mo_coeff is not None:
            mo_coeff = np.asarray(mo_coeff)
            mo_occ = np.asarray(dm.mo_occ)
            nmo = mo_occ.shape[-1]
            mo_coeff = mo_coeff[None] if mo_coeff.ndim == 2 else mo_coeff.reshape(-1, nao, nmo)   # (no copy of an F-ordered eigh result)
            mo_occ = mo_occ.reshape(-1, nmo)
            if mo_occ.shape[0] * 2 == nset:            # ROHF-style DM (df_jk.py:346-351)
                mo_coeff = np.vstack((mo_coeff, mo_coeff))
                mo_occ = np.vstack((np.array(mo_occ > 0, dtype=np.double), np.array(mo_occ == 2, dtype=np.double)))
            blocks = [_scaled_occupied(mo_coeff[k], mo_occ[k]) for k in range(nset)]
            nocc = np.array([b.shape[1] for b in blocks], dtype=np.int32)
            orbo = np.concatenate([b.ravel() for b in blocks]) if nocc.sum() else np.zeros(1)
            # dm == orbo orbo^T ?  One matrix-vector probe per density (r04 computed D v twice per density: with numpy's 64 BLAS
            # threads on a 256-core host that was the whole 10 ms gap between this call and the bare C call; a probe on a helper
            # thread beside the device call - tried in r05 - made the C call itself 40 ms slower: the BLAS threads spin)
            # Is dm == orbo orbo^T?  r06: the FULL-matrix probe (the every-16th-row probe of r05 missed sparse in-place
            # edits of a tagged array) runs INSIDE the C call, on the calling thread beside the queued kernels (flags bit 1): the
            # host pays nothing for it (before: 5-7 ms of BLAS start-up per call on a 256-thread host), and a tag that does not
            # describe its matrix gets J recomputed from the matrix before the call returns (PAMD_df_last_mismatch says so)
            flags = 2
        vj = pinned_empty(dms.shape) if with_j else None
        vk = pinned_empty(dms.shape) if with_k else None

        def call(fl):
            _check(load().PAMD_df_get_jk(
                self._h, dms.ctypes.data_as(_c.c_void_p), orbo.ctypes.data_as(_c.c_void_p) if orbo is not None else None,
                nocc.ctypes.data_as(_c.c_void_p) if nocc is not None else None, _c.c_int(nset), _c.c_int(nao), _c.c_int(hermi),
                _c.c_int(int(with_j)), _c.c_int(int(with_k)), _c.c_int(fl),
                vj.ctypes.data_as(_c.c_void_p) if with_j else None, vk.ctypes.data_as(_c.c_void_p) if with_k else None))
        call(flags)
        load().PAMD_df_last_mismatch.restype = _c.c_double
        self._last_mismatch = float(load().PAMD_df_last_mismatch(self._h)) if flags else 0.0
        self._last_fused = bool(flags) and self._last_mismatch <= 1e-10
        return (vj.reshape(shape) if with_j else None), (vk.reshape(shape) if with_k else None)
