"""Coupled-perturbed Hartree-Fock / Kohn-Sham equations for a closed-shell reference (field-independent basis).

In the place of ``pyscf/scf/cphf.py`` (``solve`` :29-51, ``solve_nos1`` :53-87): first-order orbital coefficients
mo1[a, i] (virtual x occupied, MO basis) of a one-electron perturbation h1[a, i] = (C_vir^T h^(1) C_occ),

    (e_a - e_i) mo1_ai + fvind(mo1)_ai = -h1_ai ,

with ``fvind(mo1) = C_vir^T vind(2 (C_vir mo1 C_occ^T + h.c.)) C_occ`` from ``mf.gen_response(hermi=1)`` - every product is
one symmetric first-order density through the device J/K + XC-kernel path.  The preconditioned system
(1 + fvind / (e_a - e_i)) mo1 = -h1 / (e_a - e_i) is solved by GMRES (the reference uses its own Krylov routine,
``lib.krylov``); several perturbations are solved one after the other.
"""
import numpy as np

from ..lib import tag_array
import scipy.sparse.linalg


def gen_vind(mf, mo_coeff=None, mo_occ=None):
    """fvind(mo1[..., nvir, nocc]) -> same shape: the response of the Fock matrix in the virtual-occupied block
    (the closure ``fx`` of pyscf/hessian/rhf.py:449-461 / prop/polarizability/rhf.py:58-70)."""
    if mo_coeff is None: mo_coeff = mf.mo_coeff
    if mo_occ is None: mo_occ = mf.mo_occ
    mo_coeff, mo_occ = np.asarray(mo_coeff), np.asarray(mo_occ)
    orbo, orbv = mo_coeff[:, mo_occ > 0], mo_coeff[:, mo_occ == 0]
    vind = mf.gen_response(mo_coeff, mo_occ, singlet=None, hermi=1)

    def fx(mo1):
        mo1 = np.asarray(mo1)
        shape = mo1.shape
        m = mo1.reshape(-1, orbv.shape[1], orbo.shape[1])
        lefts = np.matmul(orbv, m * 2)                            # * 2: double occupancy; d1_k = left_k orbo^T (rank nocc)
        d1 = np.matmul(lefts, orbo.T)
        v1 = vind(tag_array(d1 + d1.transpose(0, 2, 1), lowrank=([orbo] * len(lefts), list(lefts), True)))
        return np.matmul(orbv.T, np.matmul(v1, orbo)).reshape(shape)
    return fx


def solve(fvind, mo_energy, mo_occ, h1, s1=None, max_cycle=50, tol=1e-9, hermi=False, verbose=None, level_shift=0):
    """-> (mo1, None) like the reference's solve_nos1; h1 is (nvir, nocc) or (nset, nvir, nocc)."""
    if s1 is not None:
        raise NotImplementedError('CPHF with a first-order overlap (field- or geometry-dependent basis functions)')
    mo_energy, mo_occ = np.asarray(mo_energy), np.asarray(mo_occ)
    h1 = np.asarray(h1, dtype=np.float64)
    e_ai = 1.0 / (mo_energy[mo_occ == 0][:, None] + level_shift - mo_energy[mo_occ > 0][None, :])
    shape = h1.shape
    hs = h1.reshape(-1, *e_ai.shape)
    out = np.empty_like(hs)

    def matvec(x):
        x = x.reshape(e_ai.shape)
        v = fvind(x)
        if level_shift:
            v = v - x * level_shift
        return (x + v * e_ai).ravel()
    op = scipy.sparse.linalg.LinearOperator((e_ai.size, e_ai.size), matvec=matvec, dtype=np.float64)
    for k, h in enumerate(hs):
        b = (-h * e_ai).ravel()
        x, info = scipy.sparse.linalg.gmres(op, b, x0=b.copy(), rtol=tol, atol=0.0, restart=max_cycle, maxiter=3)
        if info != 0:
            raise RuntimeError('CPHF: GMRES did not reach %.1e (info = %d)' % (tol, info))
        out[k] = x.reshape(e_ai.shape)
    return out.reshape(shape), None


kernel = solve
