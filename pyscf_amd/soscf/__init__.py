"""Second-order SCF for closed-shell DF-RHF / DF-RKS: ``mf.newton()``.

In the place of ``pyscf/soscf/newton_ah.py`` (``gen_g_hop_rhf`` :49-114, ``kernel`` :506-626, ``newton`` :1034-1090).  The
reference's co-iterative augmented-Hessian scheme is replaced by the plain augmented-Hessian Newton method: each macro
iteration builds the Fock matrix once (``get_veff``: device J/K + XC), then finds the lowest eigenvector of

    [[0, g^T], [g, H]]        g = 2 F_vo,   H x = 2 [F_vv x - x F_oo + C_v^T vind(2 (C_v x C_o^T + h.c.)) C_o]

by Davidson iterations whose matrix-vector product is one ``mf.gen_response(singlet=None, hermi=1)`` call (one extra J/K
with a single symmetric matrix on the general-DM branch + ``nr_rks_fxc``), rotates the orbitals by exp(kappa) with the
step limited to ``max_stepsize``, and stops on the orbital gradient.  The converged orbitals are made canonical by one
diagonalisation of the final Fock matrix, so the object can be used wherever a DIIS-converged one can (gradients, TDDFT).
"""
import numpy as np

from ..lib import tag_array
import scipy.linalg


def gen_g_hop_rhf(mf, mo_coeff, mo_occ, fock_ao):
    """(g, h_op, h_diag) over the (nvir, nocc) rotation parameters (newton_ah.py:49-114)."""
    occ, vir = mo_occ > 0, mo_occ == 0
    orbo, orbv = mo_coeff[:, occ], mo_coeff[:, vir]
    fock = mo_coeff.T.dot(fock_ao).dot(mo_coeff)
    foo, fvv = fock[np.ix_(occ, occ)], fock[np.ix_(vir, vir)]
    g = fock[np.ix_(vir, occ)] * 2
    h_diag = (fvv.diagonal()[:, None] - foo.diagonal()[None, :]) * 2
    vind = mf.gen_response(mo_coeff, mo_occ, singlet=None, hermi=1)

    def h_op(x):
        x = x.reshape(g.shape)
        left = orbv.dot(x * 2)                             # * 2: double occupancy; d1 = left orbo^T has rank nocc
        d1 = left.dot(orbo.T)
        v1 = vind(tag_array(d1 + d1.T, lowrank=([orbo], [left], True)))
        return ((fvv.dot(x) - x.dot(foo) + orbv.T.dot(v1).dot(orbo)) * 2).ravel()
    return g.ravel(), h_op, h_diag.ravel()


def _augmented_hessian_step(g, h_op, h_diag, tol, max_cycle):
    """Lowest eigenvector (1, x) of [[0, g^T], [g, H]] -> x; Davidson in the space spanned by (1, 0) and the corrections."""
    n = g.size
    vs = [np.concatenate(([1.0], np.zeros(n)))]
    precond = -g / np.maximum(h_diag, 1e-2)
    v1 = np.concatenate(([0.0], precond))
    v1 /= np.linalg.norm(v1)
    vs.append(v1)

    def apply(v):
        return np.concatenate(([g.dot(v[1:])], g * v[0] + (h_op(v[1:]) if np.any(v[1:]) else 0.0)))
    avs = [apply(v) for v in vs]
    nhop = 1
    w = u = None
    for _ in range(max_cycle):
        vm, am = np.array(vs), np.array(avs)
        hsub = vm.dot(am.T)
        w_all, u_all = np.linalg.eigh((hsub + hsub.T) * .5)
        # follow the lowest root with a sizeable weight on the reference component
        k = next((i for i in range(len(w_all)) if abs(u_all[:, i].dot(vm[:, 0])) > 0.1), 0)
        w, u = w_all[k], u_all[:, k]
        vec = u.dot(vm)
        r = u.dot(am) - w * vec
        if np.linalg.norm(r) < tol:
            break
        d = np.concatenate(([1.0], h_diag)) - w
        d[np.abs(d) < 1e-2] = 1e-2
        t = r / d
        for _pass in range(2):
            t -= vm.T.dot(vm.dot(t))
        nt = np.linalg.norm(t)
        if nt < 1e-10:
            break
        t /= nt
        vs.append(t)
        avs.append(apply(t))
        nhop += 1
    vec = u.dot(np.array(vs))
    return vec[1:] / vec[0], w, nhop


def _canonicalize(mo_coeff, mo_occ, fock):
    """Diagonalise the Fock matrix inside the occupied and inside the virtual space (hf.py canonicalize :1095-1118)."""
    mo_coeff = np.array(mo_coeff)
    mo_energy = np.empty(mo_coeff.shape[1])
    for idx in (np.where(mo_occ > 0)[0], np.where(mo_occ == 0)[0]):
        c = mo_coeff[:, idx]
        e, u = np.linalg.eigh(c.T.dot(fock).dot(c))
        mo_coeff[:, idx] = c.dot(u)
        mo_energy[idx] = e
    return mo_coeff, mo_energy


class NewtonSCF:
    """``mf.newton()``: same attributes as the wrapped object after ``kernel()`` (mo_coeff, mo_energy, mo_occ, e_tot)."""
    max_cycle = 50
    max_stepsize = 0.2
    ah_conv_tol = 1e-3            # relative to the gradient norm (tightened as the gradient falls)
    ah_max_cycle = 30
    conv_tol_grad = None
    max_reoccupations = 5         # Aufbau re-assignments at non-Aufbau stationary points

    def __init__(self, mf):
        mo_occ = getattr(mf, 'mo_occ', None)
        if hasattr(mf, 'nelec') and np.ndim(mo_occ) == 2:
            raise NotImplementedError('second-order SCF is built for closed-shell references')
        self._scf = mf
        self.mol = mf.mol
        self.conv_tol = mf.conv_tol
        self.converged = False
        self.cycles = 0
        self.hessian_products = 0

    def __getattr__(self, name):                # everything else (get_veff, gen_response, with_df, xc, grids ...) is the SCF's
        return getattr(self._scf, name)

    def kernel(self, mo_coeff=None, mo_occ=None, dm0=None):
        mf = self._scf
        mol = self.mol
        h1e, s1e = mf.get_hcore(mol), mf.get_ovlp(mol)
        if mo_coeff is None and mf.mo_coeff is not None and np.ndim(mf.mo_coeff) == 2:
            mo_coeff, mo_occ = mf.mo_coeff, mf.mo_occ
        if mo_coeff is None:
            if dm0 is None:
                dm0 = mf.get_init_guess(mol, mf.init_guess)
            vhf = mf.get_veff(mol, dm0)
            e, mo_coeff = mf.eig(h1e + np.asarray(vhf), s1e)
            mo_occ = mf.get_occ(e, mo_coeff)
        mo_coeff, mo_occ = np.array(mo_coeff), np.asarray(mo_occ)
        if np.any((mo_occ > 0) & (mo_occ < 2)):
            raise NotImplementedError('second-order SCF is built for closed-shell references')
        tol_g = self.conv_tol_grad or np.sqrt(self.conv_tol)
        e_last = None
        nswap = 0
        for cycle in range(self.max_cycle):
            dm = mf.make_rdm1(mo_coeff, mo_occ)
            vhf = mf.get_veff(mol, dm)
            e_tot = mf.energy_tot(dm, h1e, vhf)
            fock = h1e + np.asarray(vhf)
            self.cycles = cycle + 1
            g, h_op, h_diag = gen_g_hop_rhf(mf, mo_coeff, mo_occ, fock)
            gnorm = np.linalg.norm(g)
            mf._log('macro iter %d  E = %.12f  |g| = %.3e', cycle, e_tot, gnorm)
            if gnorm < tol_g and (e_last is None or abs(e_tot - e_last) < self.conv_tol):
                # stationary point: is it the Aufbau one?  A rotation between orbitals of different symmetry has zero
                # gradient, so a wrongly occupied orbital can only be exchanged through the occupations - from the
                # pseudo-canonical orbital energies, as the reference does every cycle (newton_ah.py:573-579); here only
                # at a stationary point, where those energies mean something
                c_can, e_can = _canonicalize(mo_coeff, mo_occ, fock)
                new_occ = mf.get_occ(e_can, c_can)
                if np.any(new_occ != mo_occ) and nswap < self.max_reoccupations:
                    mf._log('stationary point is not Aufbau (E = %.12f): occupations re-assigned', e_tot)
                    mo_coeff, mo_occ = c_can, new_occ
                    nswap += 1
                    e_last = None
                    continue
                self.converged = True
                break
            e_last = e_tot
            x, _w, nhop = _augmented_hessian_step(g, h_op, h_diag, max(self.ah_conv_tol * gnorm, 1e-9), self.ah_max_cycle)
            self.hessian_products += nhop
            big = np.abs(x).max()
            if big > self.max_stepsize:
                x *= self.max_stepsize / big
            nocc = int((mo_occ > 0).sum())
            nmo = mo_coeff.shape[1]
            occ_idx, vir_idx = np.where(mo_occ > 0)[0], np.where(mo_occ == 0)[0]
            kappa = np.zeros((nmo, nmo))
            xm = x.reshape(nmo - nocc, nocc)
            kappa[np.ix_(vir_idx, occ_idx)] = xm
            kappa[np.ix_(occ_idx, vir_idx)] = -xm.T
            mo_coeff = mo_coeff.dot(scipy.linalg.expm(kappa))
        # canonical orbitals of the final Fock matrix within the occupied / virtual spaces
        dm = mf.make_rdm1(mo_coeff, mo_occ)
        vhf = mf.get_veff(mol, dm)
        fock = h1e + np.asarray(vhf)
        mo_coeff, mo_energy = _canonicalize(mo_coeff, mo_occ, fock)
        order = np.argsort(mo_energy, kind='stable')
        mf.mo_coeff, mf.mo_energy, mf.mo_occ = mo_coeff[:, order], mo_energy[order], mo_occ[order]
        mf.e_tot = mf.energy_tot(dm, h1e, vhf)
        mf.converged = self.converged
        self.e_tot = mf.e_tot
        return self.e_tot

    scf = kernel

    def run(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        self.kernel()
        return self


def newton(mf):
    return NewtonSCF(mf)


def stability_rhf_internal(mf, nroots=1, tol=1e-5):
    """Lowest eigenvalue(s) of the closed-shell orbital Hessian (real RHF -> real RHF rotations; the 'internal' analysis of
    pyscf/scf/stability.py:116-180) at the orbitals held by ``mf``: (eigenvalues, stable).  A stationary point reached
    with the wrong occupations, or a symmetry-broken minimum elsewhere, shows as a negative eigenvalue."""
    from ..tdscf import _davidson
    mo_coeff, mo_occ = np.asarray(mf.mo_coeff), np.asarray(mf.mo_occ)
    dm = mf.make_rdm1(mo_coeff, mo_occ)
    fock = mf.get_hcore() + np.asarray(mf.get_veff(mf.mol, dm))
    g, h_op, h_diag = gen_g_hop_rhf(mf, mo_coeff, mo_occ, fock)
    w, _x, _conv = _davidson(lambda xs: np.array([h_op(x) for x in xs]), h_diag, min(nroots, h_diag.size), tol, 100, 40,
                             symmetric=True)
    return w, bool(w[0] > -1e-5)
