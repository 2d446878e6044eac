"""Response of the Fock matrix to a first-order density matrix: ``mf.gen_response()``.

Mirror of ``pyscf/scf/_response_functions.py`` (``_gen_rhf_response`` :29-171, ``_gen_uhf_response`` :174-247): the
function ``vind(dm1)`` that CPHF, SOSCF and TDDFT drivers iterate on,

    closed shell   v1 = [J(dm1)] - 1/2 K_x(dm1) + f_xc : dm1          (orbital Hessian / singlet TDDFT)
                   v1 =          - 1/2 K_x(dm1) + f_xc^triplet : dm1  (triplet TDDFT)
    open shell     v1_s = [J(dm1_a + dm1_b)] - K_x(dm1_s) + sum_t f_xc^{st} : dm1_t

with K_x the exact exchange of the functional (hyb K, or its range-separated combination: the four branches of
``dft/rks.py:108-127``).  Every term is a call into the device path: ``DF.get_jk`` with several non-symmetric matrices
(``df_jk.py:302-327``, general-DM branch of the MFMA kernels) and ``NumInt.nr_rks_fxc / nr_rks_fxc_st / nr_uks_fxc``.
hermi = 2 (anti-symmetric dm1) keeps only the exchange, as in the reference.
"""
import numpy as np


def _coulomb_exchange(mf, dm1, hermi, with_j, omega, alpha, hyb, lowrank=None):
    """(J or None, exact-exchange matrix already scaled) for dm1 of any leading shape.  lowrank: the factors of the trial
    densities (tag of the caller's dm1, see df_jk._vk_lowrank), handed on to get_jk."""
    mol = mf.mol
    h = 0 if hermi == 2 else hermi
    if lowrank is not None:
        from ..lib import tag_array
        dm1 = tag_array(dm1, lowrank=lowrank)
    vj = None
    if hyb == 0 and (omega == 0 or alpha == 0):            # no exact exchange at all
        if with_j:
            vj = mf.get_jk(mol, dm1, h, with_k=False)[0]
        return vj, None
    if omega == 0:                                          # global hybrid / Hartree-Fock
        vj, vk = mf.get_jk(mol, dm1, h, with_j=with_j)
        return vj, vk * hyb
    if with_j:
        vj = mf.get_jk(mol, dm1, h, with_k=False)[0]
    if alpha == 0:                                          # short-range exchange only: erfc-attenuated tensor
        vk = mf.get_jk(mol, dm1, h, with_j=False, omega=-omega)[1] * hyb
    elif hyb == 0:                                          # long-range exchange only
        vk = mf.get_jk(mol, dm1, h, with_j=False, omega=omega)[1] * alpha
    else:                                                   # hyb K_full + (alpha - hyb) K_LR
        vk = (mf.get_jk(mol, dm1, h, with_j=False)[1] * hyb +
              mf.get_jk(mol, dm1, h, with_j=False, omega=omega)[1] * (alpha - hyb))
    return vj, vk


def _xc_coefficients(mf):
    """(numint or None, omega, alpha, hyb): Hartree-Fock counts as hyb = alpha = 1 without a grid term."""
    ni = getattr(mf, '_numint', None)
    if ni is None or not hasattr(mf, 'xc'):
        return None, 0.0, 1.0, 1.0
    omega, alpha, hyb = ni.rsh_and_hybrid_coeff(mf.xc, spin=mf.mol.spin)
    return ni, omega, alpha, hyb


def gen_rhf_response(mf, mo_coeff=None, mo_occ=None, singlet=None, hermi=0, max_memory=None):
    """vind(dm1) for a closed-shell RHF / RKS object; dm1 (nao, nao) or (nset, nao, nao), TOTAL first-order density.
    singlet None: orbital Hessian / CPHF; True / False: singlet / triplet TDDFT kernel (_response_functions.py:29-171)."""
    if mo_coeff is None: mo_coeff = mf.mo_coeff
    if mo_occ is None: mo_occ = mf.mo_occ
    mo_coeff, mo_occ = np.asarray(mo_coeff), np.asarray(mo_occ)
    if mo_coeff.ndim != 2:
        raise TypeError('gen_rhf_response needs restricted orbitals')
    mol = mf.mol
    ni, omega, alpha, hyb = _xc_coefficients(mf)
    from ..lib import tag_array
    dm0 = tag_array((mo_coeff * mo_occ).dot(mo_coeff.T), mo_coeff=mo_coeff, mo_occ=mo_occ)
    triplet = singlet is not None and not singlet

    def retag(d, scale=1.0, lowrank=None):
        if lowrank is None:
            return d * scale if scale != 1.0 else d
        lefts, rights, sym = lowrank
        return tag_array(d * scale if scale != 1.0 else d,
                         lowrank=(lefts, [r * scale for r in rights] if scale != 1.0 else rights, sym))

    def vind(dm1):
        lowrank = getattr(dm1, 'lowrank', None)
        dm1 = np.asarray(dm1)
        v1 = np.zeros_like(dm1, dtype=np.float64)
        if ni is not None and hermi != 2:
            if triplet:                      # nr_rks_fxc_st takes the alpha part of the first-order density
                v1 = v1 + ni.nr_rks_fxc_st(mol, mf.grids, mf.xc, dm0, retag(dm1, .5, lowrank), hermi=hermi, singlet=False)
            else:
                v1 = v1 + ni.nr_rks_fxc(mol, mf.grids, mf.xc, dm0, retag(dm1, 1.0, lowrank), hermi=hermi)
        vj, vk = _coulomb_exchange(mf, dm1, hermi, not triplet and hermi != 2, omega, alpha, hyb, lowrank)
        if vj is not None:
            v1 = v1 + vj
        if vk is not None:
            v1 = v1 - .5 * vk
        return v1
    return vind


def gen_uhf_response(mf, mo_coeff=None, mo_occ=None, with_j=True, hermi=0, max_memory=None):
    """vind(dm1) for UHF / UKS; dm1 = (dm1_alpha, dm1_beta), each (nao, nao) or (nset, nao, nao)
    (_response_functions.py:174-247)."""
    if mo_coeff is None: mo_coeff = mf.mo_coeff
    if mo_occ is None: mo_occ = mf.mo_occ
    mo_coeff, mo_occ = np.asarray(mo_coeff), np.asarray(mo_occ)
    if mo_coeff.ndim != 3:
        raise TypeError('gen_uhf_response needs (alpha, beta) orbitals')
    mol = mf.mol
    ni, omega, alpha, hyb = _xc_coefficients(mf)
    from ..lib import tag_array
    dm0 = tag_array(np.array([(mo_coeff[s] * mo_occ[s]).dot(mo_coeff[s].T) for s in range(2)]), mo_coeff=mo_coeff,
                    mo_occ=mo_occ)

    def vind(dm1):
        lowrank = getattr(dm1, 'lowrank', None)
        dm1 = np.asarray(dm1)
        v1 = np.zeros_like(dm1, dtype=np.float64)
        if ni is not None and hermi != 2:
            v1 = v1 + ni.nr_uks_fxc(mol, mf.grids, mf.xc, dm0,
                                    dm1 if lowrank is None else tag_array(dm1, lowrank=lowrank), hermi=hermi)
        vj, vk = _coulomb_exchange(mf, dm1, hermi, with_j and hermi != 2, omega, alpha, hyb, lowrank)
        if vj is not None:
            v1 = v1 + (vj[0] + vj[1])
        if vk is not None:
            v1 = v1 - vk
        return v1
    return vind


def gen_response(mf, *args, **kwargs):
    """Dispatch on the orbital layout: restricted closed shell -> gen_rhf_response, unrestricted -> gen_uhf_response."""
    mo_coeff = kwargs.get('mo_coeff', args[0] if args else None)
    if mo_coeff is None:
        mo_coeff = mf.mo_coeff
    if np.asarray(mo_coeff).ndim == 3:
        return gen_uhf_response(mf, *args, **kwargs)
    mo_occ = kwargs.get('mo_occ', args[1] if len(args) > 1 else None)
    if mo_occ is None:
        mo_occ = mf.mo_occ
    if np.any((np.asarray(mo_occ) > 0) & (np.asarray(mo_occ) < 2)):
        raise NotImplementedError('ROHF response (the reference routes it through the UHF form of the orbitals)')
    return gen_rhf_response(mf, *args, **kwargs)
