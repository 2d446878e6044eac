"""Linear-response excitation energies of DF-RHF / DF-RKS (closed shell) and DF-UHF / DF-UKS references: TDA and TDDFT (RPA).

Host-side drivers in the place of ``pyscf/tdscf/rhf.py`` (``TDA`` :694-806, ``TDHF`` :890-1003, ``get_ab`` :137-216,
``gen_tda_operation`` :47-104) and ``pyscf/tdscf/rks.py``; all electron-repulsion and grid work is one call of
``mf.gen_response(singlet=..., hermi=0)`` per batch of trial vectors, i.e. the device path (multi-matrix ``DF.get_jk`` on
the general-DM branch + ``nr_rks_fxc`` / ``nr_rks_fxc_st``).  For a trial amplitude X[i,a] the first-order density is
dm1 = 2 C_occ X C_vir^T and, with v = vind(dm1),

    (A X)_ia = (e_a - e_i) X_ia + (C_occ^T v   C_vir)_ia        (B X)_ia = (C_occ^T v^T C_vir)_ia

so one response call gives both blocks.  TDA diagonalises A (Davidson with the orbital-energy-difference preconditioner;
dense when the single-excitation space is small), TDDFT solves (A - B)(A + B) Z = w^2 Z, Z = X + Y.
Amplitudes are normalised like the reference: <X|X> - <Y|Y> = 1/2 (restricted) or 1 (unrestricted, ``tdscf/uhf.py``, where the
first-order spin densities are C_occ,s X_s C_vir,s^T and the amplitude vector is (X_alpha, X_beta) side by side).
"""
import numpy as np

from ..lib import tag_array

DENSE_MAX = 2000          # single-excitation dimension up to which A (and B) are formed and diagonalised directly


class TDA:
    """``tdscf.TDA(mf)``; attributes of the reference that drivers set: singlet, nstates, conv_tol, max_cycle."""

    def __init__(self, mf):
        mo_occ = np.asarray(mf.mo_occ)
        if mo_occ.ndim == 1 and np.any((mo_occ > 0) & (mo_occ < 2)):
            raise NotImplementedError('TDA / TDDFT on an ROHF reference (use the UHF form of the orbitals)')
        self.unrestricted = mo_occ.ndim == 2          # UHF / UKS reference: amplitudes (X_alpha, X_beta) side by side
        self._scf = mf
        self.mol = mf.mol
        self.singlet = True
        self.nstates = 3
        self.conv_tol = 1e-7
        self.max_cycle = 100
        self.max_space = 40
        self.batch = 64               # trial vectors per response call
        self.positive_eig_threshold = 1e-3   # roots below are dropped (near-zero modes of symmetry-broken references; tdscf/rhf.py:713)
        self.e = None
        self.xy = None
        self.converged = None

    # -- the linear operator ----------------------------------------------------------------
    def _orbitals(self):
        mf = self._scf
        occ = np.asarray(mf.mo_occ) > 0
        co, cv = np.asarray(mf.mo_coeff)[:, occ], np.asarray(mf.mo_coeff)[:, ~occ]
        de = np.asarray(mf.mo_energy)[~occ][None, :] - np.asarray(mf.mo_energy)[occ][:, None]
        return co, cv, de

    def gen_ab_operation(self):
        """-> (f, de): f(X[n, ...]) = (A X, B X), de = orbital energy differences; restricted: X is (nocc, nvir);
        unrestricted (tdscf/uhf.py:46-139): X is the flat concatenation of the alpha and beta blocks."""
        if self.unrestricted:
            return self._gen_ab_operation_uhf()
        co, cv, de = self._orbitals()
        vind = self._scf.gen_response(singlet=self.singlet, hermi=0)

        def f(xs):
            xs = np.asarray(xs).reshape(-1, *de.shape)
            ax, bx = np.empty_like(xs), np.empty_like(xs)
            for p0 in range(0, len(xs), self.batch):
                x = xs[p0:p0 + self.batch]
                # BLAS matmul chains (a three-operand einsum would run an O(n nao^2 nocc nvir) scalar loop)
                rs = 2 * np.matmul(cv, x.transpose(0, 2, 1))            # dm1_k = C_occ (2 C_vir x_k^T)^T: rank nocc
                dm1 = tag_array(np.matmul(co, rs.transpose(0, 2, 1)), lowrank=([co] * len(x), list(rs), False))
                v = vind(dm1)
                ax[p0:p0 + len(x)] = de * x + np.matmul(co.T, np.matmul(v, cv))
                bx[p0:p0 + len(x)] = np.matmul(co.T, np.matmul(v.transpose(0, 2, 1), cv))
            return ax, bx
        return f, de

    def _gen_ab_operation_uhf(self):
        mf = self._scf
        orbs = []
        for s in range(2):
            occ = np.asarray(mf.mo_occ[s]) > 0
            c, e = np.asarray(mf.mo_coeff[s]), np.asarray(mf.mo_energy[s])
            orbs.append((c[:, occ], c[:, ~occ], e[~occ][None, :] - e[occ][:, None]))
        na = orbs[0][2].size
        de = np.concatenate([o[2].ravel() for o in orbs])
        vind = mf.gen_response(hermi=0)

        def f(xs):
            xs = np.asarray(xs).reshape(-1, de.size)
            ax, bx = np.empty_like(xs), np.empty_like(xs)
            for p0 in range(0, len(xs), self.batch):
                x = xs[p0:p0 + self.batch]
                blocks = [x[:, :na].reshape(len(x), *orbs[0][2].shape), x[:, na:].reshape(len(x), *orbs[1][2].shape)]
                rs = [np.matmul(cv, xb.transpose(0, 2, 1)) for (co, cv, _), xb in zip(orbs, blocks)]
                dm1 = np.array([np.matmul(co, r.transpose(0, 2, 1)) for (co, cv, _), r in zip(orbs, rs)])
                dm1 = tag_array(dm1, lowrank=([orbs[0][0]] * len(x) + [orbs[1][0]] * len(x), list(rs[0]) + list(rs[1]), False))
                v = vind(dm1)                                     # (2, n, nao, nao)
                for out, vt in ((ax, v), (bx, v.transpose(0, 1, 3, 2))):
                    parts = [np.matmul(co.T, np.matmul(vt[s], cv)).reshape(len(x), -1) for s, (co, cv, _) in enumerate(orbs)]
                    out[p0:p0 + len(x)] = np.concatenate(parts, axis=1)
                ax[p0:p0 + len(x)] += de * x
            return ax, bx
        return f, de

    def get_ab(self):
        """Dense A, B: restricted (nocc, nvir, nocc, nvir) arrays (tdscf/rhf.py:137-216); unrestricted (nov, nov) matrices
        over the concatenated (alpha, beta) single excitations."""
        f, de = self.gen_ab_operation()
        nov = de.size
        ax, bx = f(np.eye(nov).reshape(nov, *de.shape))
        if self.unrestricted:
            return ax.reshape(nov, nov).T, bx.reshape(nov, nov).T
        a = ax.reshape(nov, nov).T.reshape(*de.shape, *de.shape)
        b = bx.reshape(nov, nov).T.reshape(*de.shape, *de.shape)
        return a, b

    def _norm(self):
        return 1.0 if self.unrestricted else .5          # <X|X> - <Y|Y> (tdscf/uhf.py:727, rhf.py:800)

    # -- solvers -------------------------------------------------------------------------------
    def kernel(self, nstates=None):
        if nstates is not None:
            self.nstates = nstates
        f, de = self.gen_ab_operation()
        nov = de.size
        n = min(self.nstates, nov)
        if nov <= DENSE_MAX:
            a = np.asarray(self.get_ab()[0]).reshape(nov, nov)
            w, v = np.linalg.eigh((a + a.T) * .5)
            keep = np.where(w > self.positive_eig_threshold)[0][:n]
            e, x = w[keep], v[:, keep].T
            self.converged = np.ones(len(e), bool)
        else:
            nextra = min(nov, n + 3)
            e, x, conv = _davidson(lambda xs: f(xs)[0].reshape(len(xs), nov), de.ravel(), nextra, self.conv_tol,
                                   self.max_cycle, self.max_space, symmetric=True)
            keep = np.where(e > self.positive_eig_threshold)[0][:n]
            e, x, self.converged = e[keep], x[keep], conv[keep]
        self.e = e
        self.xy = [(xi.reshape(de.shape) * np.sqrt(self._norm()), 0) for xi in x]
        return self.e, self.xy

    run = kernel


class TDDFT(TDA):
    """Full linear response (RPA / TDHF / TDDFT), ``tdscf.TDDFT(mf)`` / ``tdscf.TDHF(mf)`` (tdscf/rhf.py:890-1003)."""

    def kernel(self, nstates=None):
        if nstates is not None:
            self.nstates = nstates
        f, de = self.gen_ab_operation()
        nov = de.size
        n = min(self.nstates, nov)

        def apb_amb(zs):                 # ((A + B) Z, (A - B) Z)
            ax, bx = f(zs)
            return (ax + bx).reshape(len(zs), nov), (ax - bx).reshape(len(zs), nov)
        if nov <= DENSE_MAX:
            a, b = self.get_ab()
            a, b = np.asarray(a).reshape(nov, nov), np.asarray(b).reshape(nov, nov)
            w2, z = np.linalg.eig((a - b).dot(a + b))
            order = np.argsort(w2.real)
            w2, z = w2.real[order], z.real[:, order].T
            if w2[0] < -self.positive_eig_threshold ** 2:
                raise RuntimeError('TDDFT: the reference is unstable (w^2 = %.3g)' % w2[0])
            keep = np.where(w2 > self.positive_eig_threshold ** 2)[0][:n]
            w2, z = w2[keep], z[keep]
            self.converged = np.ones(len(w2), bool)
        else:
            def m(zs):
                return apb_amb(apb_amb(zs)[0])[1]
            w2, z, conv = _davidson(m, de.ravel() ** 2, min(nov, n + 3), self.conv_tol, self.max_cycle, self.max_space,
                                    symmetric=False)
            if w2[0] < -self.positive_eig_threshold ** 2:
                raise RuntimeError('TDDFT: the reference is unstable (w^2 = %.3g)' % w2[0])
            keep = np.where(w2 > self.positive_eig_threshold ** 2)[0][:n]
            w2, z, self.converged = w2[keep], z[keep], conv[keep]
        e = np.sqrt(w2)
        xy = []
        for wi, zi in zip(e, z):
            apb_z = apb_amb(zi[None])[0][0]
            xmy = apb_z / wi                 # (A + B)(X + Y) = w (X - Y)
            x, y = (zi + xmy) * .5, (zi - xmy) * .5
            norm = np.sqrt(self._norm() / abs(x.dot(x) - y.dot(y)))
            xy.append((x.reshape(de.shape) * norm, y.reshape(de.shape) * norm))
        self.e, self.xy = e, xy
        return self.e, self.xy

    run = kernel


TDHF = TDDFT
RPA = TDDFT


def _davidson(matvec, diag, nroots, tol, max_cycle, max_space, symmetric):
    """Lowest eigenpairs of the operator rows -> matvec(rows) with diagonal estimate diag.  Block Davidson with the
    (diag - theta)^-1 preconditioner; symmetric=False solves the small projected problem with a general eigensolver
    (the (A - B)(A + B) product is similar to a symmetric positive matrix, its spectrum is real)."""
    n = diag.size
    order = np.argsort(diag)
    nguess = min(n, max(nroots + 3, 2 * nroots))
    vs = np.zeros((nguess, n))
    vs[np.arange(nguess), order[:nguess]] = 1.0
    avs = matvec(vs)
    conv = np.zeros(nroots, bool)
    theta = x = None
    for _ in range(max_cycle):
        h = vs.dot(avs.T)
        if symmetric:
            w, u = np.linalg.eigh((h + h.T) * .5)
        else:
            w, u = np.linalg.eig(h)
            idx = np.argsort(w.real)
            w, u = w.real[idx], u.real[:, idx]
        theta, u = w[:nroots], u[:, :nroots]
        x = u.T.dot(vs)
        r = u.T.dot(avs) - theta[:, None] * x
        rn = np.linalg.norm(r, axis=1)
        conv = rn < tol
        if conv.all():
            break
        new = []
        for k in np.where(~conv)[0]:
            d = diag - theta[k]
            d[np.abs(d) < 1e-6] = 1e-6
            new.append(r[k] / d)
        if len(vs) + len(new) > max_space * nroots:            # restart from the current Ritz vectors
            vs = np.linalg.qr(x.T)[0].T
            avs = matvec(vs)
        t = np.array(new)
        for _pass in range(2):
            t -= t.dot(vs.T).dot(vs)
        q = np.linalg.qr(t.T)[0].T
        q -= q.dot(vs.T).dot(vs)
        keep = np.linalg.norm(q, axis=1) > 1e-8
        if not keep.any():
            break
        q = q[keep] / np.linalg.norm(q[keep], axis=1)[:, None]
        vs = np.vstack([vs, q])
        avs = np.vstack([avs, matvec(q)])
    xn = x / np.linalg.norm(x, axis=1)[:, None]
    return theta, xn, conv
