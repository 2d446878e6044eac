"""Restricted open-shell HF on top of the DF J/K path.

Mirror of ``pyscf/scf/rohf.py``: Roothaan effective Fock (:121-154), ``get_fock`` (:83-119),
``get_occ`` / ``_fill_rohf_occ`` (:157-227), ``get_grad`` (:229-260), ``make_rdm1`` (:262-290; the
(alpha, beta) densities are tagged with ONE coefficient matrix and 0/1/2 occupations - the layout the
DF K build turns into two occupied blocks, pyscf/df/df_jk.py:346-351), UHF-type ``get_veff`` and energy."""
import time

import numpy as np

from ..lib import tag_array
from . import hf


def get_roothaan_fock(focka_fockb, dma_dmb, s):
    nao = s.shape[0]
    focka, fockb = focka_fockb
    dma, dmb = dma_dmb
    fc = (focka + fockb) * .5
    pc = dmb.dot(s)
    po = (dma - dmb).dot(s)
    pv = np.eye(nao) - dma.dot(s)
    fock = pc.conj().T.dot(fc).dot(pc) * .5
    fock += po.conj().T.dot(fc).dot(po) * .5
    fock += pv.conj().T.dot(fc).dot(pv) * .5
    fock += po.conj().T.dot(fockb).dot(pc)
    fock += po.conj().T.dot(focka).dot(pv)
    fock += pv.conj().T.dot(fc).dot(pc)
    fock = fock + fock.conj().T
    return tag_array(fock, focka=focka, fockb=fockb)


class ROHF(hf.SCF):
    def __init__(self, mol):
        hf.SCF.__init__(self, mol)
        self.nelec = mol.nelec

    def get_init_guess(self, mol=None, key='minao', s1e=None):
        dm = np.asarray(hf.SCF.get_init_guess(self, mol, key, s1e))
        na, nb = self.nelec
        ne = max(na + nb, 1)
        return np.array((dm * (na / ne), dm * (nb / ne)))

    def eig(self, fock, s, x=None):
        e, c = hf.SCF.eig(self, np.asarray(fock), s, x)
        if getattr(fock, 'focka', None) is not None:
            mo_ea = np.einsum('pi,pi->i', c.conj(), fock.focka.dot(c)).real
            mo_eb = np.einsum('pi,pi->i', c.conj(), fock.fockb.dot(c)).real
            e = tag_array(e, mo_ea=mo_ea, mo_eb=mo_eb)
        return e, c

    def get_occ(self, mo_energy, mo_coeff=None):
        mo_ea = getattr(mo_energy, 'mo_ea', mo_energy)
        na, nb = self.nelec
        nocc, ncore = (na, nb) if na > nb else (nb, na)
        nopen = nocc - ncore
        e = np.asarray(mo_energy)
        mo_occ = np.zeros_like(e)
        core_sort = np.argsort(e)
        mo_occ[core_sort[:ncore]] = 2
        if nopen > 0:
            open_idx = core_sort[ncore:]
            open_sort = np.argsort(np.asarray(mo_ea)[open_idx])
            mo_occ[open_idx[open_sort[:nopen]]] = 1
        return mo_occ

    def make_rdm1(self, mo_coeff=None, mo_occ=None):
        if mo_coeff is None: mo_coeff = self.mo_coeff
        if mo_occ is None: mo_occ = self.mo_occ
        ca = mo_coeff[:, mo_occ > 0]
        cb = mo_coeff[:, mo_occ == 2]
        dm = np.array((ca.dot(ca.conj().T), cb.dot(cb.conj().T)))
        return tag_array(dm, mo_coeff=mo_coeff, mo_occ=mo_occ)

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
        if dm is None: dm = self.make_rdm1()
        if np.ndim(dm) == 2:
            dm = np.array((np.asarray(dm) * .5, np.asarray(dm) * .5))
        t0 = time.perf_counter()
        vj, vk = self.get_jk(mol, dm, hermi)
        self._log('df vj and vk: %.4f s', time.perf_counter() - t0)
        return vj[0] + vj[1] - vk

    def get_fock(self, h1e, s1e, vhf, dm, cycle=-1, diis=None, fock_last=None):
        dm = np.asarray(dm)
        if dm.ndim == 2:
            dm = np.array((dm * .5, dm * .5))
        focka, fockb = h1e + vhf[0], h1e + vhf[1]
        f = get_roothaan_fock((focka, fockb), dm, s1e)
        if abs(getattr(self, 'damp', 0)) > 1e-4 or abs(getattr(self, 'level_shift', 0)) > 1e-4:
            # pyscf/scf/rohf.py get_fock damps and level-shifts the Roothaan Fock matrix; not restated here - refuse rather
            # than run silently unshifted iterations
            raise NotImplementedError('ROHF: damp / level_shift are not implemented (set them to 0)')
        if cycle < 0 or diis is None:
            return f
        if cycle >= self.diis_start_cycle:
            f = tag_array(diis.update(s1e, dm[0] + dm[1], np.asarray(f)), focka=focka, fockb=fockb)
        return f

    def get_grad(self, mo_coeff, mo_occ, fock):
        occa, occb = mo_occ > 0, mo_occ == 2
        uniq_a = (~occa).reshape(-1, 1) & occa
        uniq_b = (~occb).reshape(-1, 1) & occb
        if getattr(fock, 'focka', None) is not None:
            focka, fockb = fock.focka, fock.fockb
        else:
            focka = fockb = fock
        fa = mo_coeff.conj().T.dot(focka).dot(mo_coeff)
        fb = mo_coeff.conj().T.dot(fockb).dot(mo_coeff)
        g = np.zeros_like(fa)
        g[uniq_a] = fa[uniq_a]
        g[uniq_b] += fb[uniq_b]
        return g[uniq_a | uniq_b]

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        if dm is None: dm = self.make_rdm1()
        if h1e is None: h1e = self.get_hcore()
        if vhf is None: vhf = self.get_veff(self.mol, dm)
        dm = np.asarray(dm)
        e1 = np.einsum('ij,ji->', h1e, dm[0] + dm[1]).real
        e_coul = (np.einsum('ij,ji->', vhf[0], dm[0]) + np.einsum('ij,ji->', vhf[1], dm[1])).real * .5
        self.scf_summary.update(e1=e1, e2=e_coul)
        return e1 + e_coul, e_coul
