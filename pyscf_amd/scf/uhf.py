"""Unrestricted HF on top of the DF J/K path.

Mirror of ``pyscf/scf/uhf.py``: ``get_veff`` (:227-300,:1059-1090: V_s = J[D_a + D_b] - K[D_s]),
``energy_elec`` (:310-340), ``get_occ`` (:381-440: lowest n_alpha / n_beta orbitals),
``make_rdm1`` (:141-170, tagged with the (2, nao, nmo) coefficients so the DF K build takes the MO
branch with two occupied blocks), ``get_grad`` (:350-370), DIIS on the stacked (alpha, beta) Fock
matrices (pyscf/scf/diis.py:40-96 handles the 3-index case by summing the per-spin errors).
The density matrices go through ``with_df.get_jk`` as one (2, nao, nao) batch - one pass over cderi."""
import time

import numpy as np

from ..lib import tag_array
from . import hf


class UHF(hf.SCF):
    def __init__(self, mol):
        hf.SCF.__init__(self, mol)
        self.nelec = mol.nelec

    def get_init_guess(self, mol=None, key='minao', s1e=None):
        dm = hf.SCF.get_init_guess(self, mol, key, s1e)
        dm = np.asarray(dm)
        na, nb = self.nelec
        # pyscf/scf/uhf.py:42-58: split the closed-shell guess; scale to the alpha / beta electron numbers
        ne = max(na + nb, 1)
        return np.array((dm * (na / ne), dm * (nb / ne)))

    def eig(self, h, s, x=None):
        ea, ca = hf.SCF.eig(self, h[0], s, x)
        eb, cb = hf.SCF.eig(self, h[1], s, x)
        return np.array((ea, eb)), np.array((ca, cb))

    def get_occ(self, mo_energy, mo_coeff=None):
        occ = np.zeros_like(mo_energy)
        for s, n in enumerate(self.nelec):
            idx = np.argsort(mo_energy[s].round(9), kind='stable')
            occ[s, idx[:n]] = 1
        return occ

    def make_rdm1(self, mo_coeff=None, mo_occ=None):
        if mo_coeff is None: mo_coeff = self.mo_coeff
        if mo_occ is None: mo_occ = self.mo_occ
        dms = []
        for s in range(2):
            c = mo_coeff[s][:, mo_occ[s] > 0]
            dms.append((c * mo_occ[s][mo_occ[s] > 0]).dot(c.conj().T))
        return tag_array(np.array(dms), mo_coeff=mo_coeff, mo_occ=mo_occ, dm_from_orbitals=True)

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
        if dm is None: dm = self.make_rdm1()
        if np.ndim(dm) == 2:
            dm = np.repeat(np.asarray(dm)[None] * .5, 2, axis=0)
        t0 = time.perf_counter()
        vj, vk = self.get_jk(mol, dm, hermi)
        self._log('df vj and vk: %.4f s', time.perf_counter() - t0)
        vj = vj[0] + vj[1]
        vhf = vj - vk
        ecoul = np.einsum('nij,ji->', np.asarray(dm), vj).real * .5
        return tag_array(vhf, ecoul=ecoul)

    def get_fock(self, h1e, s1e, vhf, dm, cycle=-1, diis=None, fock_last=None):
        """uhf.py:297-337: as the restricted form per spin; level_shift may be a pair (alpha, beta)."""
        f = h1e + vhf
        if cycle < 0 and diis is None:
            return f
        shift = self.level_shift
        shifta, shiftb = shift if isinstance(shift, (tuple, list, np.ndarray)) else (shift, shift)
        if 0 <= cycle < self.diis_start_cycle - 1 and abs(self.damp) > 1e-4 and fock_last is not None:
            f = hf.damping(f, np.asarray(fock_last), self.damp)
        if diis is not None and cycle >= self.diis_start_cycle:
            f = diis.update(s1e, dm, f)
        if abs(shifta) + abs(shiftb) > 1e-4:
            dm = np.asarray(dm)
            f = np.array((hf.level_shift(s1e, dm[0], f[0], shifta), hf.level_shift(s1e, dm[1], f[1], shiftb)))
        return f

    def get_grad(self, mo_coeff, mo_occ, fock):
        g = []
        for s in range(2):
            occ, vir = mo_occ[s] > 0, mo_occ[s] == 0
            g.append(mo_coeff[s][:, vir].conj().T.dot(fock[s].dot(mo_coeff[s][:, occ])).ravel())
        return np.hstack(g)

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        if dm is None: dm = self.make_rdm1()
        if h1e is None: h1e = self.get_hcore()
        if vhf is None: vhf = self.get_veff(self.mol, dm)
        e1 = np.einsum('ij,ji->', h1e, dm[0] + dm[1]).real
        e_coul = (np.einsum('ij,ji->', vhf[0], dm[0]) + np.einsum('ij,ji->', vhf[1], dm[1])).real * .5
        self.scf_summary.update(e1=e1, e2=e_coul)
        return e1 + e_coul, e_coul

    def spin_square(self):
        """<S^2> of the UHF determinant (pyscf/scf/uhf.py:201-224)."""
        s = self.get_ovlp()
        ca = self.mo_coeff[0][:, self.mo_occ[0] > 0]
        cb = self.mo_coeff[1][:, self.mo_occ[1] > 0]
        na, nb = ca.shape[1], cb.shape[1]
        sab = ca.T.dot(s).dot(cb)
        ssxy = (na + nb) * .5 - np.einsum('ij,ij->', sab, sab)
        ssz = (na - nb) ** 2 * .25
        return ssxy + ssz
