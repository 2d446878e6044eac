"""numpy-only clients of the host-array XC entry points (include/pyscf_amd.h: PAMD_grid_weights_host, PAMD_xc_create,
PAMD_xc_nr_rks, PAMD_xc_nr_uks) - the XC siblings of ``pyscf_amd.df.native.NativeDF``.

``NativeGrids`` is ``pyscf.dft.gen_grid.Grids`` (pyscf/dft/gen_grid.py:487-744) with the Becke partition done by the library on
host arrays (the reference hands that loop to C as well: VXCgen_grid, pyscf/lib/dft/grid_basis.c:32-101); the radial / Lebedev
/ pruning tables are host numpy as in the reference.  ``NativeNumInt`` duck-types the part of ``pyscf.dft.numint.NumInt`` that
``RKS.get_veff`` / ``UKS.get_veff`` use (``nr_rks``, ``nr_uks``, ``rsh_and_hybrid_coeff``, ``hybrid_coeff``, ``_xc_type``,
pyscf/dft/rks.py:76-131, pyscf/dft/numint.py:1074-1324): the functional string is parsed on the host (libxc.parse_xc, as the
reference does before LIBXC_eval_xc), densities reach C as orbital factors (the tag of ``make_rdm1``, or a signed
eigen-factorisation by numpy), results come back in caller-owned numpy arrays.  No torch, no device pointers in Python.

    mf = dft.RKS(mol, xc='b3lyp')
    mf.with_df = NativeDF(mol, devices=[0]); mf._numint = NativeNumInt(); mf.grids = NativeGrids(mol)
"""
import ctypes

import numpy as np

from ..df import native as _native
from . import gen_grid, libxc as _xc

_c = ctypes


def _ptr(a):
    return a.ctypes.data_as(_c.c_void_p)


class NativeGrids(gen_grid.Grids):
    """Grids with get_partition on host arrays (PAMD_grid_weights_host) - everything else inherited."""

    def __init__(self, mol, device=0):
        gen_grid.Grids.__init__(self, mol)
        self.device = device

    def get_partition(self, mol, atom_grids_tab):
        lib = _native.load()
        scheme = gen_grid.scheme_id(self.becke_scheme)
        table = None
        if callable(self.radii_adjust) and self.atomic_radii is not None:
            table = np.ascontiguousarray(self.radii_adjust(mol, self.atomic_radii), dtype=np.float64)
        atm_coords = np.ascontiguousarray(mol.atom_coords(), dtype=np.float64)
        dev = int(self.device) if not isinstance(self.device, str) else 0
        coords_all, weights_all = [], []
        for ia in range(mol.natm):
            c, vol = atom_grids_tab[mol.atom_symbol(ia)]
            c = np.ascontiguousarray(c + atm_coords[ia], dtype=np.float64)
            vol = np.ascontiguousarray(vol, dtype=np.float64)
            w = np.empty(len(vol))
            _native._check(lib.PAMD_grid_weights_host(_ptr(c), _c.c_long(len(vol)), _ptr(atm_coords), _c.c_int(mol.natm),
                                                      _ptr(table) if table is not None else None, _c.c_int(scheme), _c.c_int(ia),
                                                      _ptr(vol), _c.c_int(dev), _ptr(w)))
            coords_all.append(c)
            weights_all.append(w)
        return np.vstack(coords_all), np.hstack(weights_all)


class NativeNumInt:
    libxc = _xc
    omega = None

    def __init__(self, device=0, devices=None):
        self.device = device
        self.devices = None if devices is None else [int(d) for d in devices]     # several GPUs in this process: PAMD_xc_create_multi
        self._h = None
        self._key = None

    # -- functional properties (numint.py:2737-2800) -------------------------------------------------------------------
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

    def _parse(self, xc_code):
        hyb, fac = _xc.parse_xc(xc_code)
        if self.omega is not None and fac[_xc.F_OMEGA] != 0:
            fac = fac.copy()
            fac[_xc.F_OMEGA] = abs(float(self.omega))
        return hyb, np.ascontiguousarray(fac, dtype=np.float64)

    # -- handle ---------------------------------------------------------------------------------------------------------
    def reset(self):
        if self._h is not None:
            _native.load().PAMD_xc_destroy(self._h)
        self._h = None
        self._key = None
        return self

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    def _handle(self, mol, grids):
        if grids.coords is None:
            grids.build()
        key = (id(mol), np.asarray(mol._env).tobytes(), id(grids), getattr(grids, '_build_id', 0), grids.size)
        if self._h is None or key != self._key:
            self.reset()
            atm = np.ascontiguousarray(mol._atm, dtype=np.int32)
            bas = np.ascontiguousarray(mol._bas, dtype=np.int32)
            env = np.ascontiguousarray(mol._env, dtype=np.float64)
            coords = np.ascontiguousarray(grids.coords, dtype=np.float64)
            weights = np.ascontiguousarray(grids.weights, dtype=np.float64)
            h = _c.c_void_p()
            if self.devices is not None:
                devs = (_c.c_int * len(self.devices))(*self.devices)
                _native._check(_native.load().PAMD_xc_create_multi(_ptr(atm), _c.c_int(len(atm)), _ptr(bas), _c.c_int(len(bas)), _ptr(env),
                                                                   _c.c_int(len(env)), _ptr(coords), _ptr(weights), _c.c_long(len(weights)),
                                                                   devs, _c.c_int(len(self.devices)), _c.byref(h)))
            else:
                _native._check(_native.load().PAMD_xc_create(_ptr(atm), _c.c_int(len(atm)), _ptr(bas), _c.c_int(len(bas)), _ptr(env),
                                                             _c.c_int(len(env)), _ptr(coords), _ptr(weights), _c.c_long(len(weights)),
                                                             _c.c_int(int(self.device)), _c.byref(h)))
            self._h, self._key = h, key
        return self._h

    def plan_info(self, mol, grids, xc_code='lda'):
        """{'tiles', 'density', 'compact_GB'} of the block-sparse plan (built when needed)."""
        info = (_c.c_double * 3)()
        _native._check(_native.load().PAMD_xc_plan_info(self._handle(mol, grids), _c.c_int(int(_xc.xc_type(xc_code) == 'GGA')), info))
        return dict(tiles=int(info[0]), density=info[1], compact_GB=info[2])

    def last_timing(self, mol, grids):
        """HIP-event timings of the kernels of the last nr_rks / nr_uks inside the handle (PAMD_xc_last_timing) and the executed flops
        of its two MFMA products - what bench.py --single-process prices as `xc_path.roofline`."""
        out = (_c.c_double * 10)()
        _native._check(_native.load().PAMD_xc_last_timing(self._handle(mol, grids), out, _c.c_int(10)))
        g, ncomp, npad = out[7], out[8], out[9]
        return dict(parts=int(out[0]), ms={'ao_dot_mo': out[1], 'eval_xc': out[2], 'scale_ao': out[3], 'ao_dot_aow': out[4]},
                    flops={'ao_dot_mo': 2.0 * ncomp * g * out[5] * npad, 'ao_dot_aow': 2.0 * g * out[6]},
                    scale_bytes=8.0 * (ncomp + 1) * g * out[5])

    @staticmethod
    def _factors(dm, mo_coeff, mo_occ):
        """(orb (nao, r) C order, signs (r) | None): D = sum_i s_i c_i c_i^T - the tag's occupied orbitals scaled by sqrt(occ)
        (numint.py:2930-2994 MO branch), else the eigen-factorisation of the symmetric part (a density only sees that part)."""
        if mo_coeff is not None:
            return _native._scaled_occupied(np.asarray(mo_coeff), np.asarray(mo_occ)), None
        w, v = np.linalg.eigh((dm + dm.T) * .5)
        keep = abs(w) > 1e-14 * max(abs(w).max(), 1e-300)
        sg = np.sign(w[keep])
        return np.ascontiguousarray(v[:, keep] * np.sqrt(abs(w[keep]))), (np.ascontiguousarray(sg) if (sg < 0).any() else None)

    # -- the entry points -----------------------------------------------------------------------------------------------
    def nr_rks(self, mol, grids, xc_code, dms, relativity=0, hermi=1, max_memory=2000, verbose=None):
        """-> (nelec, excsum, vmat) with the contract of numint.nr_rks (pyscf/dft/numint.py:1074-1190)."""
        dms_arr = np.asarray(dms)
        nao = dms_arr.shape[-1]
        shape = dms_arr.shape
        dms2 = dms_arr.reshape(-1, nao, nao)
        nset = len(dms2)
        nelec, excsum = np.zeros(nset), np.zeros(nset)
        xctype = _xc.xc_type(xc_code)
        # the library writes every element of vmat: page-locked memory (copied at the PCIe rate) instead of zero-filled pageable pages
        vmat = np.zeros((nset, nao, nao)) if xctype == 'HF' else _native.pinned_empty((nset, nao, nao))
        if xctype != 'HF':
            if xctype not in ('LDA', 'GGA'):
                raise NotImplementedError('xc type %s' % xctype)
            hyb, fac = self._parse(xc_code)
            mo_coeff, mo_occ = getattr(dms, 'mo_coeff', None), getattr(dms, 'mo_occ', None)
            tagged = mo_coeff is not None and np.ndim(mo_occ) == 1 and nset == 1
            facs = [self._factors(dms2[s], mo_coeff if tagged else None, mo_occ) for s in range(nset)]
            nocc = np.array([f[0].shape[1] for f in facs], dtype=np.int32)
            orbs = np.concatenate([f[0].ravel() for f in facs]) if nocc.sum() else np.zeros(1)
            signs = None
            if any(f[1] is not None for f in facs):
                signs = np.concatenate([f[1] if f[1] is not None else np.ones(f[0].shape[1]) for f in facs])
            _native._check(_native.load().PAMD_xc_nr_rks(
                self._handle(mol, grids), _ptr(fac), _c.c_int(int(xctype == 'GGA')), _c.c_int(nset), _ptr(orbs), _ptr(nocc),
                _ptr(signs) if signs is not None else None, _ptr(nelec), _ptr(excsum), _ptr(vmat)))
        if dms_arr.ndim == 2:
            return nelec[0], excsum[0], vmat[0]
        return nelec, excsum, vmat.reshape(shape)

    def nr_uks(self, mol, grids, xc_code, dms, relativity=0, hermi=1, max_memory=2000, verbose=None):
        """-> (nelec[2], excsum, vmat[2]) with the contract of numint.nr_uks (pyscf/dft/numint.py:1192-1324)."""
        dms_arr = np.asarray(dms)
        nao = dms_arr.shape[-1]
        assert dms_arr.shape == (2, nao, nao), 'nr_uks: one (alpha, beta) pair'
        nelec, exc = np.zeros(2), np.zeros(1)
        xctype = _xc.xc_type(xc_code)
        vmat = np.zeros((2, nao, nao)) if xctype == 'HF' else _native.pinned_empty((2, nao, nao))
        if xctype != 'HF':
            if xctype not in ('LDA', 'GGA'):
                raise NotImplementedError('xc type %s' % xctype)
            hyb, fac = self._parse(xc_code)
            mo_coeff, mo_occ = getattr(dms, 'mo_coeff', None), getattr(dms, 'mo_occ', None)
            tagged = mo_coeff is not None and np.ndim(mo_occ) == 2
            facs = [self._factors(dms_arr[s], np.asarray(mo_coeff)[s] if tagged else None, np.asarray(mo_occ)[s] if tagged else None)
                    for s in range(2)]
            nocc = np.array([f[0].shape[1] for f in facs], dtype=np.int32)
            orbs = np.concatenate([f[0].ravel() for f in facs] + [np.zeros(1)])
            signs = None
            if any(f[1] is not None for f in facs):
                signs = np.concatenate([f[1] if f[1] is not None else np.ones(f[0].shape[1]) for f in facs])
            _native._check(_native.load().PAMD_xc_nr_uks(
                self._handle(mol, grids), _ptr(fac), _c.c_int(int(xctype == 'GGA')), _ptr(orbs), _ptr(nocc),
                _ptr(signs) if signs is not None else None, _ptr(nelec), _ptr(exc), _ptr(vmat)))
        return nelec, exc[0], vmat

    def nr_vxc(self, mol, grids, xc_code, dms, spin=0, relativity=0, hermi=1, max_memory=2000, verbose=None):
        return (self.nr_uks if spin else self.nr_rks)(mol, grids, xc_code, dms, relativity, hermi, max_memory, verbose)
