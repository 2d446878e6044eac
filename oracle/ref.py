"""ORACLE - TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by pyscf_amd/).

numpy restatement of the reference's density-fitting algorithm on top of the C integral
oracle (oracle/cint_oracle.c):

* ``cholesky_eri``  <- pyscf/df/incore.py:129-220  (j2c Cholesky + trsm, eig fallback :263-270)
* ``get_jk``        <- pyscf/df/df_jk.py:280-413   (J via packed-tril two-pass product, K via
                       per-row unpack + symm half transform + X^T X; general-DM branch :382-408)
* ``pack_tril`` / ``unpack_tril`` <- pyscf/lib/numpy_helper.py:328-466
* ``fp``            <- pyscf/lib/misc.py:1359-1363
* ``rhf_kernel``    <- pyscf/scf/hf.py:49-241 (+ CDIIS pyscf/scf/diis.py:40-96,
                       pyscf/lib/diis.py:225-290)

Parity status: pinned against the reference's golden vectors G1-G7 in
tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.linalg

_here = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _here])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_here, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _lib = ctypes.CDLL(so)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fp(a):
    """lib.fp fingerprint: sum(cos(arange(n)) * a.ravel())  (pyscf/lib/misc.py:1359-1363)."""
    a = np.asarray(a)
    return np.dot(np.cos(np.arange(a.size)), a.ravel())


def pack_tril(mat):
    mat = np.asarray(mat)
    n = mat.shape[-1]
    idx = np.tril_indices(n)
    return mat[..., idx[0], idx[1]]


def unpack_tril(tril, filltriu=1):
    tril = np.asarray(tril)
    npair = tril.shape[-1]
    n = int((np.sqrt(8 * npair + 1) - 1) / 2)
    if filltriu and tril.ndim == 2 and tril.dtype == np.float64 and tril.shape[0] * n * n > 1 << 22:
        t = np.ascontiguousarray(tril)
        out = np.empty((t.shape[0], n, n))
        lib().oracle_unpack_tril(_p(t), _p(out), ctypes.c_int(t.shape[0]), ctypes.c_int(n))
        return out
    idx = np.tril_indices(n)
    out = np.zeros(tril.shape[:-1] + (n, n))
    out[..., idx[0], idx[1]] = tril
    if filltriu:
        out[..., idx[1], idx[0]] = tril
    return out


# ----------------------------------------------------------------------------- integrals
def _tables(mol):
    return (np.ascontiguousarray(mol._atm, np.int32), np.ascontiguousarray(mol._bas, np.int32),
            np.ascontiguousarray(mol._env, np.float64))


def int1e(mol, kind):
    """kind: 'ovlp' | 'kin' | 'nuc'  (int1e_ovlp_sph / int1e_kin_sph / int1e_nuc_sph)."""
    atm, bas, env = _tables(mol)
    n = mol.nao_nr()
    out = np.zeros((n, n))
    lib().oracle_int1e(ctypes.c_int({'ovlp': 0, 'kin': 1, 'nuc': 2}[kind]), _p(out), _p(atm),
                       ctypes.c_int(len(atm)), _p(bas), ctypes.c_int(len(bas)), _p(env))
    return out


def int2c2e(auxmol, omega=0.0):
    atm, bas, env = _tables(auxmol)
    n = auxmol.nao_nr()
    out = np.zeros((n, n))
    lib().oracle_set_omega(ctypes.c_double(omega))
    lib().oracle_int2c2e(_p(out), _p(atm), ctypes.c_int(len(atm)), _p(bas), ctypes.c_int(0),
                         ctypes.c_int(len(bas)), _p(env))
    lib().oracle_set_omega(ctypes.c_double(0.0))
    return out


def int3c2e(mol, auxmol, omega=0.0):
    """(naux, nao, nao) s1 tensor == reference's int3c2e_sph viewed as [k][i][j]."""
    from pyscf_amd.gto import conc_env
    atm, bas, env = conc_env(mol._atm, mol._bas, mol._env, auxmol._atm, auxmol._bas, auxmol._env)
    atm = np.ascontiguousarray(atm, np.int32)
    bas = np.ascontiguousarray(bas, np.int32)
    env = np.ascontiguousarray(env)
    nao, naux = mol.nao_nr(), auxmol.nao_nr()
    out = np.zeros((naux, nao, nao))
    lib().oracle_set_omega(ctypes.c_double(omega))
    lib().oracle_int3c2e(_p(out), _p(atm), ctypes.c_int(len(atm)), _p(bas),
                         ctypes.c_int(mol.nbas), ctypes.c_int(auxmol.nbas), _p(env))
    lib().oracle_set_omega(ctypes.c_double(0.0))
    return out


def ao_loc(mol):
    """Function offset of every AO shell (segmented or general contraction): mol.ao_loc_nr()."""
    bas = np.asarray(mol._bas)
    return np.concatenate([[0], np.cumsum((2 * bas[:, 1] + 1) * bas[:, 3])]).astype(int)


def int3c2e_slab(mol, auxmol, ish0, ish1, omega=0.0, out=None):
    """Packed raw integrals (Q|pq) for the AO rows p of shells [ish0, ish1), all q <= p, all aux Q:
    (naux, p1(p1+1)/2 - p0(p0+1)/2), the column slab [pq0, pq1) of the reference's s2ij tensor
    (pyscf/df/incore.py:178-199)."""
    from pyscf_amd.gto import conc_env
    atm, bas, env = conc_env(mol._atm, mol._bas, mol._env, auxmol._atm, auxmol._bas, auxmol._env)
    atm = np.ascontiguousarray(atm, np.int32)
    bas = np.ascontiguousarray(bas, np.int32)
    env = np.ascontiguousarray(env)
    loc = ao_loc(mol)
    p0, p1 = int(loc[ish0]), int(loc[ish1])
    ncol = p1 * (p1 + 1) // 2 - p0 * (p0 + 1) // 2
    naux = auxmol.nao_nr()
    if out is None:
        out = np.zeros((naux, ncol))
    assert out.shape == (naux, ncol) and out.flags.c_contiguous
    lib().oracle_set_omega(ctypes.c_double(omega))
    lib().oracle_int3c2e_slab(_p(out), ctypes.c_long(ncol), _p(atm), ctypes.c_int(len(atm)), _p(bas),
                              ctypes.c_int(mol.nbas), ctypes.c_int(auxmol.nbas), _p(env), ctypes.c_int(ish0),
                              ctypes.c_int(ish1))
    lib().oracle_set_omega(ctypes.c_double(0.0))
    return out


def int3c2e_block(mol, auxmol, ish0, ish1, jsh0, jsh1, omega=0.0):
    """Raw integrals (Q|pq) for p in AO shells [ish0, ish1), q in AO shells [jsh0, jsh1), all aux Q: (naux, np, nq), no
    symmetry packing (the rectangular counterpart of int3c2e_slab)."""
    from pyscf_amd.gto import conc_env
    atm, bas, env = conc_env(mol._atm, mol._bas, mol._env, auxmol._atm, auxmol._bas, auxmol._env)
    atm = np.ascontiguousarray(atm, np.int32)
    bas = np.ascontiguousarray(bas, np.int32)
    env = np.ascontiguousarray(env)
    loc = ao_loc(mol)
    out = np.zeros((auxmol.nao_nr(), int(loc[ish1] - loc[ish0]), int(loc[jsh1] - loc[jsh0])))
    lib().oracle_set_omega(ctypes.c_double(omega))
    lib().oracle_int3c2e_block(_p(out), _p(atm), ctypes.c_int(len(atm)), _p(bas), ctypes.c_int(mol.nbas),
                               ctypes.c_int(auxmol.nbas), _p(env), ctypes.c_int(ish0), ctypes.c_int(ish1), ctypes.c_int(jsh0),
                               ctypes.c_int(jsh1))
    lib().oracle_set_omega(ctypes.c_double(0.0))
    return out


def int2e(mol, omega=0.0):
    atm, bas, env = _tables(mol)
    n = mol.nao_nr()
    out = np.zeros((n, n, n, n))
    lib().oracle_set_omega(ctypes.c_double(omega))
    lib().oracle_int2e(_p(out), _p(atm), ctypes.c_int(len(atm)), _p(bas),
                       ctypes.c_int(len(bas)), _p(env))
    lib().oracle_set_omega(ctypes.c_double(0.0))
    return out


# ----------------------------------------------------------------------------- DF tensor
LINEAR_DEP_THR = 1e-7  # pyscf/df/incore.py:33


def cholesky_eri(mol, auxmol, lindep=LINEAR_DEP_THR, omega=0.0):
    """cderi (naux, nao_pair), B = L^-1 (Q|pq)   (pyscf/df/incore.py:129-220)."""
    j2c = int2c2e(auxmol, omega)
    j3c = pack_tril(int3c2e(mol, auxmol, omega))         # (naux, nao_pair), s2ij
    try:
        low = scipy.linalg.cholesky(j2c, lower=True)
        return scipy.linalg.solve_triangular(low, j3c, lower=True, check_finite=False)
    except scipy.linalg.LinAlgError:
        w, v = scipy.linalg.eigh(j2c)
        mask = w > lindep
        v = v[:, mask] / np.sqrt(w[mask])
        return v.T.dot(j3c)


# ----------------------------------------------------------------------------- J/K
def get_jk(cderi, dm, hermi=1, with_j=True, with_k=True, mo_coeff=None, mo_occ=None,
           blockdim=240):
    """Restatement of pyscf/df/df_jk.py:329-411 in numpy."""
    dms = np.asarray(dm)
    shape = dms.shape
    nao = shape[-1]
    dms = dms.reshape(-1, nao, nao)
    nset = len(dms)
    naux = cderi.shape[0]
    vj = 0
    vk = np.zeros_like(dms)
    if with_j:
        idx = np.arange(nao)
        dmtril = pack_tril(dms + dms.transpose(0, 2, 1))
        dmtril[:, idx * (idx + 1) // 2 + idx] *= .5
    orbo = None
    if with_k and mo_coeff is not None:
        mo_coeff = np.asarray(mo_coeff).reshape(-1, nao, np.asarray(mo_occ).shape[-1])
        mo_occ = np.asarray(mo_occ).reshape(-1, mo_coeff.shape[-1])
        orbo = [mo_coeff[k][:, mo_occ[k] > 0] * np.sqrt(mo_occ[k][mo_occ[k] > 0])
                for k in range(nset)]
    for b0 in range(0, naux, blockdim):
        eri1 = cderi[b0:b0 + blockdim]
        if with_j:
            vj = vj + dmtril.dot(eri1.T).dot(eri1)
        if with_k:
            full = unpack_tril(eri1)                      # (blk, nao, nao)
            for k in range(nset):
                if orbo is not None:
                    # dsymm per aux row (nr_ao2mo.c:399-419): buf1[L,i,p] = sum_q B_L[p,q] orbo[q,i]
                    buf1 = np.matmul(np.ascontiguousarray(orbo[k].T)[None], full).reshape(-1, nao)
                    vk[k] += buf1.T.dot(buf1)       # lib.dot / NPdgemm (df_jk.py:380)
                else:
                    buf1 = np.matmul(np.ascontiguousarray(dms[k].T)[None], full)   # [p][k][i]
                    vk[k] += buf1.reshape(-1, nao).T.dot(full.reshape(-1, nao))
    if with_j:
        vj = unpack_tril(vj, 1).reshape(shape)
    else:
        vj = None
    vk = vk.reshape(shape) if with_k else None
    return vj, vk


def get_jk_rows_parallel(cderi, dm, mo_coeff, mo_occ, nthreads=None, blockdim=240):
    """MO-branch J/K (pyscf/df/df_jk.py:329-381) with the reference's own parallel structure, for the CPU baseline of
    bench.py: ``AO2MOnr_e2_drv`` runs its OpenMP loop over the aux rows of a block, every thread unpacking one row and
    calling a single-threaded ``dsymm`` on it (pyscf/lib/ao2mo/nr_ao2mo.c:1253-1265, :399-419, :1016-1031); the J dot
    products (df_jk.py:367) and the final ``lib.dot(buf1.T, buf1)`` (df_jk.py:380 -> NPdgemm) are multi-threaded BLAS.
    Returns (vj, vk, flops) for one closed-shell density."""
    import concurrent.futures
    from threadpoolctl import threadpool_limits
    dm = np.asarray(dm)
    nao = dm.shape[-1]
    mo_occ = np.asarray(mo_occ)
    orbo = np.asarray(mo_coeff)[:, mo_occ > 0] * np.sqrt(mo_occ[mo_occ > 0])
    nocc = orbo.shape[1]
    orbo_t = np.ascontiguousarray(orbo.T)
    nthreads = nthreads or os.cpu_count()
    naux = cderi.shape[0]
    blockdim = max(blockdim, nthreads)          # at least one row per thread in a block
    idx = np.arange(nao)
    dmtril = pack_tril(dm + dm.T)
    dmtril[idx * (idx + 1) // 2 + idx] *= .5
    vj = np.zeros(cderi.shape[1])
    vk = np.zeros((nao, nao))

    def rows(args):
        full, buf1, r0, r1 = args
        for r in range(r0, r1):
            # buf1[r] (nocc, nao) = orbo^T . B_L: the reference calls dsymm on the symmetric B_L (nr_ao2mo.c:399-419); the
            # same product and flop count through numpy's dgemm, which - unlike scipy's f2py BLAS wrappers - releases the
            # GIL, so the worker threads really run side by side
            np.dot(orbo_t, full[r], out=buf1[r])
        return r1 - r0

    import time
    tm = {'j': 0.0, 'unpack': 0.0, 'dsymm': 0.0, 'dgemm': 0.0}
    with concurrent.futures.ThreadPoolExecutor(nthreads) as pool:
        for b0 in range(0, naux, blockdim):
            eri1 = np.ascontiguousarray(cderi[b0:b0 + blockdim])
            nb = eri1.shape[0]
            t0 = time.perf_counter()
            vj += dmtril.dot(eri1.T).dot(eri1)                       # threaded BLAS
            t1 = time.perf_counter()
            full = unpack_tril(eri1)                                  # C, OpenMP over the rows (NPdunpack_tril per row)
            t2 = time.perf_counter()
            buf1 = np.empty((nb, nocc, nao))
            per = max(1, -(-nb // nthreads))
            with threadpool_limits(limits=1, user_api='blas'):
                list(pool.map(rows, [(full, buf1, r, min(r + per, nb)) for r in range(0, nb, per)]))
            t3 = time.perf_counter()
            b2 = buf1.reshape(-1, nao)
            vk += b2.T.dot(b2)                                        # threaded dgemm (lib.dot, df_jk.py:380)
            t4 = time.perf_counter()
            tm['j'] += t1 - t0
            tm['unpack'] += t2 - t1
            tm['dsymm'] += t3 - t2
            tm['dgemm'] += t4 - t3
    flops = 2.0 * naux * nao * nao * nocc * 2 + 4.0 * naux * cderi.shape[1]
    get_jk_rows_parallel.last_phases = {k: round(v, 3) for k, v in tm.items()}
    return unpack_tril(vj[None], 1)[0], vk, flops


def get_jk_exact(eri, dm):
    """4-centre J/K from the full ERI tensor (config-1 plumbing reference)."""
    dm = np.asarray(dm)
    vj = np.einsum('ijkl,...lk->...ij', eri, dm)
    vk = np.einsum('ikjl,...kl->...ij', eri, dm)   # K_ij = (ik|jl) D_kl
    return vj, vk


# ----------------------------------------------------------------------------- SCF driver
class CDIIS:
    """Commutator DIIS (pyscf/scf/diis.py:40-96; pyscf/lib/diis.py:225-290), space 8."""

    def __init__(self, space=8):
        self.space = space
        self.fs, self.es = [], []

    def update(self, s, d, f, x_orth):
        sdf = s.dot(d).dot(f)
        err = x_orth.T.dot(sdf.T - sdf).dot(x_orth)      # C^T (FDS - SDF) C
        self.fs.append(f.copy())
        self.es.append(err.ravel())
        if len(self.fs) > self.space:
            self.fs.pop(0)
            self.es.pop(0)
        n = len(self.fs)
        h = np.zeros((n + 1, n + 1))
        h[0, 1:] = h[1:, 0] = 1
        for i in range(n):
            for j in range(n):
                h[i + 1, j + 1] = np.dot(self.es[i], self.es[j])
        g = np.zeros(n + 1)
        g[0] = 1
        w, v = scipy.linalg.eigh(h)
        if np.any(abs(w) < 1e-14):
            idx = abs(w) > 1e-14
            c = np.dot(v[:, idx] * (1. / w[idx]), np.dot(v[:, idx].T, g))
        else:
            c = np.linalg.solve(h, g)
        return sum(ci * fi for ci, fi in zip(c[1:], self.fs))


def rhf_kernel(mol, get_veff, conv_tol=1e-10, max_cycle=60, dm0=None, h1e=None, s1e=None,
               verbose=False, e2_fn=None):
    """Minimal RHF loop with the reference's semantics (hf.py:49-241): core-Hamiltonian
    ('1e') initial guess unless dm0 is given, CDIIS from cycle 1, canonical
    orthogonalisation x_orth with threshold 1e-6 (hf.py:1363-1379), E = Tr(hD)+1/2 Tr(VD)+E_nuc.
    ``get_veff(dm, mo_coeff, mo_occ) -> vhf``."""
    if h1e is None:
        h1e = int1e(mol, 'kin') + int1e(mol, 'nuc')
    if s1e is None:
        s1e = int1e(mol, 'ovlp')
    enuc = mol.energy_nuc()
    nocc = mol.nelectron // 2
    w, v = scipy.linalg.eigh(s1e)
    keep = w > 1e-6
    x_orth = v[:, keep] / np.sqrt(w[keep])

    def eig(f):
        e, c = scipy.linalg.eigh(x_orth.T.dot(f).dot(x_orth))
        return e, x_orth.dot(c)

    def make_rdm1(c, occ):
        co = c[:, occ > 0]
        return (co * occ[occ > 0]).dot(co.T)

    mo_occ = np.zeros(x_orth.shape[1])
    mo_occ[:nocc] = 2
    if dm0 is None:
        e, c = eig(h1e)
        dm = make_rdm1(c, mo_occ)
    else:
        dm = dm0
        c = None
    def etot(dm, vhf):
        # HF: E = Tr(hD) + 1/2 Tr(VD) + Enuc;  KS: the caller supplies the two-electron energy
        e2 = .5 * np.einsum('ij,ji', vhf, dm) if e2_fn is None else e2_fn()
        return np.einsum('ij,ji', h1e, dm) + e2 + enuc

    vhf = get_veff(dm, c, mo_occ if c is not None else None)
    e_tot = etot(dm, vhf)
    diis = CDIIS()
    conv = False
    for cycle in range(max_cycle):
        f = h1e + vhf
        if cycle >= 1:
            f = diis.update(s1e, dm, f, x_orth)
        e, c = eig(f)
        dm = make_rdm1(c, mo_occ)
        vhf = get_veff(dm, c, mo_occ)
        e_last = e_tot
        e_tot = etot(dm, vhf)
        f = h1e + vhf
        g = c[:, mo_occ == 0].T.dot(f).dot(c[:, mo_occ > 0]) * 2
        ng = np.linalg.norm(g)
        if verbose:
            print('cycle %d E=%.12f dE=%.3g |g|=%.3g' % (cycle + 1, e_tot, e_tot - e_last, ng), flush=True)
        if abs(e_tot - e_last) < conv_tol and ng < np.sqrt(conv_tol):
            conv = True
            break
    return conv, e_tot, e, c, mo_occ, dm


def uhf_kernel(mol, cderi, nelec, conv_tol=1e-10, max_cycle=80, mo0=None):
    """Minimal DF-UHF loop with the oracle pieces (pyscf/scf/uhf.py semantics: V_s = J[Da+Db] - K[Ds]).
    mo0 = (Ca, Cb) optional starting orbitals (to follow a given SCF state); default core-Hamiltonian guess."""
    h1e = int1e(mol, 'kin') + int1e(mol, 'nuc')
    s1e = int1e(mol, 'ovlp')
    enuc = mol.energy_nuc()
    w, v = scipy.linalg.eigh(s1e)
    x = v[:, w > 1e-6] / np.sqrt(w[w > 1e-6])

    def eig(f):
        e, c = scipy.linalg.eigh(x.T.dot(f).dot(x))
        return e, x.dot(c)
    e0, c0 = eig(h1e)
    cs = [c0, c0] if mo0 is None else [np.asarray(mo0[0]), np.asarray(mo0[1])]
    diis_f, diis_e = [], []
    e_tot = 0
    for cycle in range(max_cycle):
        dms = np.array([cs[s][:, :nelec[s]].dot(cs[s][:, :nelec[s]].T) for s in range(2)])
        vj, vk = get_jk(cderi, dms, 1)
        vhf = vj[0] + vj[1] - vk
        e_last = e_tot
        e_tot = np.einsum('ij,ji', h1e, dms[0] + dms[1]) + .5 * sum(np.einsum('ij,ji', vhf[s], dms[s]) for s in range(2)) + enuc
        f = h1e + vhf
        err = np.hstack([x.T.dot(f[s].dot(dms[s]).dot(s1e) - s1e.dot(dms[s]).dot(f[s])).dot(x).ravel() for s in range(2)])
        if abs(e_tot - e_last) < conv_tol and np.linalg.norm(err) < 1e-5:
            return True, e_tot
        diis_f.append(f); diis_e.append(err)
        diis_f, diis_e = diis_f[-8:], diis_e[-8:]
        n = len(diis_f)
        h = np.zeros((n + 1, n + 1)); h[0, 1:] = h[1:, 0] = 1
        for i in range(n):
            for j in range(n):
                h[i + 1, j + 1] = diis_e[i].dot(diis_e[j])
        g = np.zeros(n + 1); g[0] = 1
        ww, vv = scipy.linalg.eigh(h)
        idx = abs(ww) > 1e-14
        c = np.dot(vv[:, idx] * (1. / ww[idx]), vv[:, idx].T.dot(g))
        f = sum(ci * fi for ci, fi in zip(c[1:], diis_f))
        cs = [eig(f[s])[1] for s in range(2)]
    return False, e_tot
