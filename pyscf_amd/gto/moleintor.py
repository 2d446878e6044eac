"""Host driver of the on-device integral engine.

Mirrors the role of ``pyscf/gto/moleintor.py`` (``getints3c`` :530-601, ``getints2c``
:475-528): takes the libcint-format ``_atm/_bas/_env`` tables and launches the gfx950
kernel family (pyscf_amd/csrc/int3c2e_kernel.h) through the C ABI.  Host work here is
one-time table preparation (numpy): shell-pair lists per (l_i >= l_j) class, primitive-pair
records {zeta, P, K_ab c_i c_j, P-A}, aux shells grouped by l_aux, cart->sph matrices.

Conventions: real spherical functions, p = (x,y,z), l>=2 m=-l..l,
Cartesians ordered lx descending then ly descending, coefficients in env already carry the
radial normalisation (pyscf/gto/mole.py:1006,1020-1029).
"""
import ctypes
import math

import numpy as np

from .. import lib as _lib_mod
from . import mole as _mole

EXPCUTOFF = 60.0          # primitive-pair screening: drop exp(-mu R^2) < e^-60 (libcint default)
LMAX_AO = 4
LMAX_AUX = 6          # i fitting shells (def2-universal-jkfit of the 3d metals) with AO shells up to f


# --------------------------------------------------------------------------- cart -> sph
def _cart_list(l):
    return [(x, y, l - x - y) for x in range(l, -1, -1) for y in range(l - x, -1, -1)]


def c2s_matrix(l):
    """Real solid harmonics in Cartesian monomials, normalised on the unit sphere.
    Standard closed form (Helgaker/Jorgensen/Olsen eq. 6.4.47); ordering of the reference:
    pyscf/lib/parameters.py:69-77, pyscf/symm/sph.py:24-56."""
    carts = _cart_list(l)
    idx = {c: i for i, c in enumerate(carts)}
    out = np.zeros((2 * l + 1, len(carts)))
    for m in range(-l, l + 1):
        am = abs(m)
        N = (1.0 / (2 ** am * math.factorial(l)) *
             math.sqrt(2.0 * math.factorial(l + am) * math.factorial(l - am) / (2.0 if m == 0 else 1.0)))
        N *= math.sqrt((2 * l + 1) / (4 * math.pi))
        row = ({1: 0, -1: 1, 0: 2}[m] if l == 1 else m + l)
        for t in range((l - am) // 2 + 1):
            for u in range(t + 1):
                kmax = am // 2 if m >= 0 else (am - 1) // 2
                for k in range(kmax + 1):
                    twov = 2 * k if m >= 0 else 2 * k + 1
                    c = ((-1) ** (t + k) * 0.25 ** t * math.comb(l, t) * math.comb(l - t, am + t) *
                         math.comb(t, u) * math.comb(am, twov))
                    ly = 2 * u + twov
                    lx = 2 * t + am - ly
                    lz = l - 2 * t - am
                    if lx < 0 or lz < 0:
                        continue
                    out[row, idx[(lx, ly, lz)]] += N * c
    return out


# --------------------------------------------------------------------------- shell tables
class _Shells:
    """Segmented (nctr == 1) shells of a bas table; general contractions are split, which
    keeps the AO order (contraction index outer, m inner: moleintor.py:805-821)."""

    def __init__(self, atm, bas, env):
        l, xyz, exps, coefs, ao0, atom = [], [], [], [], [], []
        off = 0
        for b in bas:
            ia, ll, nprim, nctr = int(b[_mole.ATOM_OF]), int(b[_mole.ANG_OF]), int(b[_mole.NPRIM_OF]), int(b[_mole.NCTR_OF])
            pe, pc = int(b[_mole.PTR_EXP]), int(b[_mole.PTR_COEFF])
            r = env[atm[ia, _mole.PTR_COORD]:atm[ia, _mole.PTR_COORD] + 3]
            e = env[pe:pe + nprim]
            c = env[pc:pc + nprim * nctr].reshape(nctr, nprim)
            for k in range(nctr):
                nz = c[k] != 0
                l.append(ll)
                xyz.append(r)
                exps.append(e[nz].copy())
                coefs.append(c[k][nz].copy())
                ao0.append(off)
                atom.append(ia)
                off += 2 * ll + 1
        self.l = np.array(l, dtype=np.int32)
        self.xyz = np.array(xyz, dtype=np.float64).reshape(-1, 3)
        self.exps = exps
        self.coefs = coefs
        self.ao0 = np.array(ao0, dtype=np.int32)
        self.atom = np.array(atom, dtype=np.int32)
        self.nao = off
        self.n = len(l)


def _dev(arr, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device)


class _Args(ctypes.Structure):
    _fields_ = [('pair_ish', ctypes.c_void_p), ('pair_jsh', ctypes.c_void_p),
                ('pair_pp0', ctypes.c_void_p), ('pair_npp', ctypes.c_void_p),
                ('pp', ctypes.c_void_p), ('shell_xyz', ctypes.c_void_p),
                ('shell_ao0', ctypes.c_void_p),
                ('aux_f0', ctypes.c_void_p), ('aux_xyz', ctypes.c_void_p),
                ('aux_exp', ctypes.c_void_p), ('aux_coef', ctypes.c_void_p),
                ('naux_cls', ctypes.c_int), ('npk', ctypes.c_int),
                ('rys_table', ctypes.c_void_p),
                ('c2s', ctypes.c_void_p), ('c2s_off', ctypes.c_void_p),
                ('T', ctypes.c_void_p), ('ldT', ctypes.c_long), ('row_offset', ctypes.c_long),
                ('tril', ctypes.c_int), ('npairs', ctypes.c_int), ('omega', ctypes.c_double)]


class _GradArgs(ctypes.Structure):
    """Mirror of PAMD_int3c2e_grad_args (include/pyscf_amd.h)."""
    _fields_ = [('base', _Args), ('pp_ab', ctypes.c_void_p), ('shell_atom', ctypes.c_void_p),
                ('aux_atom', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('nrep', ctypes.c_int),
                ('natm', ctypes.c_int), ('aux_response', ctypes.c_int)]


class _AuxClass:
    def __init__(self, shells, l, device, sel=None):
        idx = [i for i in range(shells.n) if shells.l[i] == l and (sel is None or sel[i])]
        self.l = l
        self.n = len(idx)
        if self.n == 0:
            return
        npk = max(len(shells.exps[i]) for i in idx)
        ex = np.ones((self.n, npk))
        co = np.zeros((self.n, npk))
        for j, i in enumerate(idx):
            k = len(shells.exps[i])
            ex[j, :k] = shells.exps[i]
            co[j, :k] = shells.coefs[i]
        self.npk = npk
        self.f0 = _dev(shells.ao0[idx], device)
        self.atom = _dev(np.asarray(shells.atom)[idx].astype(np.int32), device) if hasattr(shells, 'atom') else None
        self.xyz = _dev(shells.xyz[idx], device)
        self.exp = _dev(ex, device)
        self.coef = _dev(co, device)


class _PairClass:
    """All shell pairs (a, b) with l_a = li >= l_b = lj, sorted by the row shell (the one with
    the larger AO offset) so that an AO-row slab is a contiguous sub-range."""

    def __init__(self, sa, li, lj, device):
        self.li, self.lj = li, lj
        self.n = 0
        ia_all = np.nonzero(sa.l == li)[0]
        ib_all = np.nonzero(sa.l == lj)[0]
        if len(ia_all) == 0 or len(ib_all) == 0:
            return
        nprim = np.array([len(e) for e in sa.exps])
        ish_l, jsh_l, npp_l, recs, abs_l = [], [], [], [], []
        # vectorised over groups of shells with equal primitive counts (rectangular arrays)
        for na in np.unique(nprim[ia_all]):
            A = ia_all[nprim[ia_all] == na]
            EA = np.array([sa.exps[i] for i in A])            # [nA, na]
            CA = np.array([sa.coefs[i] for i in A])
            XA = sa.xyz[A]
            for nb in np.unique(nprim[ib_all]):
                B = ib_all[nprim[ib_all] == nb]
                EB = np.array([sa.exps[i] for i in B])
                CB = np.array([sa.coefs[i] for i in B])
                XB = sa.xyz[B]
                pmask = (A[:, None] >= B[None, :]) if li == lj else np.ones((len(A), len(B)), bool)
                rab = XA[:, None, :] - XB[None, :, :]
                r2 = np.einsum('abx,abx->ab', rab, rab)
                zeta = EA[:, None, :, None] + EB[None, :, None, :]                 # [nA, nB, na, nb]
                arg = EA[:, None, :, None] * EB[None, :, None, :] / zeta * r2[:, :, None, None]
                keep = (arg < EXPCUTOFF) & pmask[:, :, None, None]
                cnt = keep.sum(axis=(2, 3))
                pa, pb = np.nonzero(cnt)
                if len(pa) == 0:
                    continue
                ka, kb, kp, kq = np.nonzero(keep)                                 # grouped by (a, b)
                z = zeta[ka, kb, kp, kq]
                wa = EA[ka, kp] / z
                P = wa[:, None] * XA[ka] + (1 - wa)[:, None] * XB[kb]
                rec = np.empty((len(z), 8))
                rec[:, 0] = z
                rec[:, 1:4] = P
                rec[:, 4] = np.exp(-arg[ka, kb, kp, kq]) * CA[ka, kp] * CB[kb, kq]
                rec[:, 5:8] = P - XA[ka]
                recs.append(rec)
                abs_l.append(np.stack([EA[ka, kp], EB[kb, kq]], axis=1))
                ish_l.append(A[pa])
                jsh_l.append(B[pb])
                npp_l.append(cnt[pa, pb])
        if not recs:
            return
        ish = np.concatenate(ish_l).astype(np.int32)
        jsh = np.concatenate(jsh_l).astype(np.int32)
        npp = np.concatenate(npp_l).astype(np.int32)
        pp0 = (np.cumsum(npp) - npp).astype(np.int32)
        rowshell = np.maximum(ish, jsh)
        order = np.argsort(rowshell, kind='stable')
        self.n = len(order)
        self.rowshell = rowshell[order]
        self.ish = _dev(ish[order], device)
        self.jsh = _dev(jsh[order], device)
        self.pp0 = _dev(pp0[order], device)
        self.npp = _dev(npp[order], device)
        self.pp = _dev(np.vstack(recs), device)
        self._pp_ab_host = np.vstack(abs_l)       # primitive exponents (alpha_i, alpha_j) per record: gradients only
        self._pp_ab = None
        self.device = device

    @property
    def pp_ab(self):
        if self._pp_ab is None:
            self._pp_ab = _dev(self._pp_ab_host, self.device)
        return self._pp_ab

    def subrange(self, sh0, sh1):
        """[i0, i1) of pairs whose row shell lies in [sh0, sh1)."""
        return (int(np.searchsorted(self.rowshell, sh0, 'left')),
                int(np.searchsorted(self.rowshell, sh1, 'left')))


class IntEngine:
    """Device-resident tables for (ij|k), (P|Q) over mol / auxmol."""

    def __init__(self, mol, auxmol, device, omega=0.0):
        import torch
        self.torch = torch
        self.device = device
        # omega > 0: erf(omega r12)/r12; omega < 0: the short-range complement erfc(|omega| r12)/r12, as libcint reads
        # env[PTR_RANGE_OMEGA] (pyscf/gto/mole.py:76-84) - generated as Coulomb minus long-range, two kernel passes
        self.omega = float(omega)
        self.lib = _lib_mod.load_library()
        self.ao = _Shells(mol._atm, mol._bas, mol._env)
        self.aux = _Shells(auxmol._atm, auxmol._bas, auxmol._env) if auxmol is not None else None
        if self.ao.l.max() > LMAX_AO:
            raise NotImplementedError('AO angular momentum > %d' % LMAX_AO)
        if self.aux is not None and self.aux.l.max() > LMAX_AUX:
            raise NotImplementedError('aux angular momentum > %d' % LMAX_AUX)
        if self.aux is not None and self.aux.l.max() > 5 and self.ao.l.max() > 3:
            raise NotImplementedError('i fitting shells are instantiated with AO shells up to f only')
        n = self.lib.PAMD_rys_table_len()
        self.rys = torch.empty(n, dtype=torch.float64, device=device)
        if torch.device(device).type == 'cuda':
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib_mod.check(self.lib.PAMD_rys_table_upload(ctypes.c_void_p(self.rys.data_ptr()), st))
        mats = [c2s_matrix(l) for l in range(max(LMAX_AO, LMAX_AUX) + 1)]
        off = np.cumsum([0] + [m.size for m in mats])[:-1].astype(np.int32)
        self.c2s = _dev(np.concatenate([m.ravel() for m in mats]), device)
        self.c2s_off = _dev(off, device)
        self.ao_xyz = _dev(self.ao.xyz, device)
        self.ao_ao0 = _dev(self.ao.ao0, device)
        self._pair_classes = None
        self._aux_classes = None

    # -- lazily built tables -------------------------------------------------------------
    def pair_classes(self):
        if self._pair_classes is None:
            self._pair_classes = []
            for li in range(int(self.ao.l.max()) + 1):
                for lj in range(li + 1):
                    pc = _PairClass(self.ao, li, lj, self.device)
                    if pc.n:
                        self._pair_classes.append(pc)
        return self._pair_classes

    def aux_classes(self):
        if self._aux_classes is None:
            self._aux_classes = [c for c in (_AuxClass(self.aux, l, self.device)
                                             for l in range(int(self.aux.l.max()) + 1)) if c.n]
        return self._aux_classes

    def _launch(self, pc, i0, i1, ac, T, ldT, row_offset, tril, shell_xyz, shell_ao0):
        if i1 <= i0:
            return
        a = _Args()
        self._fill_args(a, pc, i0, i1, ac, T, ldT, row_offset, tril, shell_xyz, shell_ao0)
        st = ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        _lib_mod.check(self.lib.PAMD_int3c2e_class(ctypes.c_int(pc.li), ctypes.c_int(pc.lj),
                                                   ctypes.c_int(ac.l), ctypes.byref(a), st))

    def grad_launch(self, pc, ac, Z, ldZ, tril, shell_xyz, shell_ao0, shell_atom, grad, aux_response, i0=0, i1=None, row_offset=0):
        """grad[rep][atom][3] += sum Z[row(pq) - row_offset][Q] d(pq|Q)/dR for the pairs [i0, i1) of one (pair class, aux class):
        PAMD_int3c2e_grad_class.  A slab of Z (packed rows from row_offset on) goes with the pair sub-range of its row shells."""
        if i1 is None:
            i1 = pc.n
        if i1 <= i0 or ac.n == 0:
            return
        g = _GradArgs()
        self._fill_args(g.base, pc, i0, i1, ac, Z, ldZ, row_offset, tril, shell_xyz, shell_ao0)
        g.pp_ab = pc.pp_ab.data_ptr()
        g.shell_atom = shell_atom.data_ptr()
        g.aux_atom = ac.atom.data_ptr()
        g.grad = grad.data_ptr()
        g.nrep, g.natm = grad.shape[0], grad.shape[1]
        g.aux_response = int(bool(aux_response))
        st = ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        _lib_mod.check(self.lib.PAMD_int3c2e_grad_class(ctypes.c_int(pc.li), ctypes.c_int(pc.lj),
                                                        ctypes.c_int(ac.l), ctypes.byref(g), st))

    def _fill_args(self, a, pc, i0, i1, ac, T, ldT, row_offset, tril, shell_xyz, shell_ao0):
        a.pair_ish = pc.ish.data_ptr() + 4 * i0
        a.pair_jsh = pc.jsh.data_ptr() + 4 * i0
        a.pair_pp0 = pc.pp0.data_ptr() + 4 * i0
        a.pair_npp = pc.npp.data_ptr() + 4 * i0
        a.pp = pc.pp.data_ptr()
        a.shell_xyz = shell_xyz.data_ptr()
        a.shell_ao0 = shell_ao0.data_ptr()
        a.aux_f0 = ac.f0.data_ptr()
        a.aux_xyz = ac.xyz.data_ptr()
        a.aux_exp = ac.exp.data_ptr()
        a.aux_coef = ac.coef.data_ptr()
        a.naux_cls = ac.n
        a.npk = ac.npk
        a.rys_table = self.rys.data_ptr()
        a.c2s = self.c2s.data_ptr()
        a.c2s_off = self.c2s_off.data_ptr()
        a.T = T.data_ptr()
        a.ldT = ldT
        a.row_offset = row_offset
        a.tril = tril
        a.npairs = i1 - i0
        a.omega = getattr(self, '_omega_override', self.omega)
        if a.omega < 0:
            raise NotImplementedError('short-range (omega < 0) integrals are generated by int3c2e_slab / int2c2e as Coulomb '
                                      'minus long-range; the gradient kernels have no such second pass')

    # -- integrals ------------------------------------------------------------------------
    def slab_rows(self, sh0, sh1):
        """packed-tril row range [r0, r1) covered by AO row shells [sh0, sh1)."""
        p0 = int(self.ao.ao0[sh0])
        p1 = int(self.ao.ao0[sh1]) if sh1 < self.ao.n else self.ao.nao
        return p0 * (p0 + 1) // 2, p1 * (p1 + 1) // 2

    def _short_range(self, fn, *args, out=None):
        """Coulomb pass into out, long-range pass (|omega|) into a scratch tensor, difference in place."""
        try:
            self._omega_override = 0.0
            full = fn(*args, out=out)
            self._omega_override = -self.omega
            full -= fn(*args, out=None)
        finally:
            del self._omega_override
        return full

    def int3c2e_slab(self, sh0, sh1, out=None):
        """T[pq - r0][Q] = (pq|Q) for the AO rows of shells [sh0, sh1); device tensor."""
        if self.omega < 0 and not hasattr(self, '_omega_override'):
            return self._short_range(self.int3c2e_slab, sh0, sh1, out=out)
        r0, r1 = self.slab_rows(sh0, sh1)
        naux = self.aux.nao
        if out is None:
            out = self.torch.zeros((r1 - r0, naux), dtype=self.torch.float64, device=self.device)
        else:
            out = out[:r1 - r0]
            out.zero_()
        for pc in self.pair_classes():
            i0, i1 = pc.subrange(sh0, sh1)
            for ac in self.aux_classes():
                self._launch(pc, i0, i1, ac, out, naux, r0, 1, self.ao_xyz, self.ao_ao0)
        return out

    def pair_classes_2c(self):
        """[(P, unit s)] pair classes over the aux shells: the 2-centre (P|Q) family."""
        dummy = _Shells.__new__(_Shells)
        dummy.coefs = [np.ones(1) / c2s_matrix(0)[0, 0]]
        out = []
        for li in range(int(self.aux.l.max()) + 1):
            pc = _PairClass2c(self.aux, li, dummy, self.device)
            if pc.n:
                out.append(pc)
        return out

    def int2c2e(self, out=None):
        """(P|Q) over the aux basis, (naux, naux) device tensor (GTOint2c analogue)."""
        if self.omega < 0 and not hasattr(self, '_omega_override'):
            return self._short_range(self.int2c2e)
        torch = self.torch
        naux = self.aux.nao
        out = torch.zeros((naux, naux), dtype=torch.float64, device=self.device)
        dummy = _Shells.__new__(_Shells)
        dummy.l = np.zeros(1, np.int32)
        dummy.xyz = np.zeros((1, 3))
        dummy.exps = [np.zeros(1)]
        dummy.coefs = [np.ones(1) / c2s_matrix(0)[0, 0]]     # cancels the s-type angular factor
        dummy.ao0 = np.zeros(1, np.int32)
        dummy.n = 1
        aux_xyz = _dev(self.aux.xyz, self.device)
        aux_ao0 = _dev(self.aux.ao0, self.device)
        for li in range(int(self.aux.l.max()) + 1):
            pc = _PairClass2c(self.aux, li, dummy, self.device)
            if not pc.n:
                continue
            for ac in self.aux_classes():
                self._launch(pc, 0, pc.n, ac, out, naux, 0, 0, aux_xyz, aux_ao0)
        return out


class _PairClass2c(_PairClass):
    """(P, unit s function at the same centre): gives the 2-centre integrals (P|Q)."""

    def __init__(self, shells, li, dummy, device):
        ia = np.nonzero(shells.l == li)[0]
        self.li, self.lj = li, 0
        self.n = len(ia)
        if self.n == 0:
            return
        recs, pp0, npp, abs_l = [], [], [], []
        nrec = 0
        for a in ia:
            e, c = shells.exps[a], shells.coefs[a] * dummy.coefs[0][0]
            rec = np.zeros((len(e), 8))
            rec[:, 0] = e
            rec[:, 1:4] = shells.xyz[a]
            rec[:, 4] = c
            recs.append(rec)
            abs_l.append(np.stack([e, np.zeros(len(e))], axis=1))
            pp0.append(nrec)
            npp.append(len(e))
            nrec += len(e)
        self.rowshell = np.asarray(ia)
        self.ish = _dev(ia.astype(np.int32), device)
        self.jsh = _dev(ia.astype(np.int32), device)      # same centre: A - B = 0
        self.pp0 = _dev(np.array(pp0, np.int32), device)
        self.npp = _dev(np.array(npp, np.int32), device)
        self.pp = _dev(np.vstack(recs), device)
        self._pp_ab_host = np.vstack(abs_l)
        self._pp_ab = None
        self.device = device


_ENGINE_CACHE = {}


def mol_fingerprint(mol):
    """Content key of a molecule's integral tables (_atm, _bas, _env): caches keyed on it survive neither an in-place
    ``mol.build(atom=...)`` nor the reuse of an ``id()`` by a new object."""
    import hashlib
    if mol is None:
        return None
    h = hashlib.sha1()
    for a in (mol._atm, mol._bas, mol._env):
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def get_engine(mol, auxmol, device, omega=0.0):
    """IntEngine cached per (mol tables, auxmol tables, device, omega): the shell-pair tables are the expensive host part.
    The key is the content of the integral tables, not the object identity."""
    import torch
    devkey = str(torch.device(device))
    key = (mol_fingerprint(mol), mol_fingerprint(auxmol), devkey + '|%.6f' % omega)
    eng = _ENGINE_CACHE.get(key)
    if eng is not None:
        return eng
    # an engine built for the same mol with another aux basis can lend its AO pair tables
    eng = IntEngine(mol, auxmol, device, omega)
    for (k0, k1, k2), e in list(_ENGINE_CACHE.items()):
        if k0 == key[0] and k2.split('|')[0] == devkey and e._pair_classes is not None:
            eng._pair_classes = e._pair_classes
            break
    if len(_ENGINE_CACHE) > 8:
        _ENGINE_CACHE.clear()
    _ENGINE_CACHE[key] = eng
    return eng


def getints(name, atm, bas, env, shls_slice=None, hermi=0, aosym='s1'):
    raise NotImplementedError('generic Mole.intor: use pyscf_amd.gto.moleintor.IntEngine / scf.hf.get_hcore')
