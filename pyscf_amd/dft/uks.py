"""Unrestricted Kohn-Sham driver (pyscf/dft/uks.py:36-120 get_veff, :123-160 energy_elec) on top of the
DF-UHF driver: V_s = J[D_a + D_b] - hyb K[D_s] + Vxc_s from ``NumInt.nr_uks``."""
import time

import numpy as np

from ..lib import tag_array
from ..scf import uhf
from . import gen_grid, numint


class UKS(uhf.UHF):
    def __init__(self, mol, xc='LDA,VWN'):
        uhf.UHF.__init__(self, mol)
        self.xc = xc
        self.grids = gen_grid.Grids(mol)
        self._numint = numint.NumInt()

    @property
    def omega(self):
        """Range-separation parameter override (KohnShamDFT.omega, pyscf/dft/rks.py:445-455): lives on the NumInt object."""
        return self._numint.omega

    @omega.setter
    def omega(self, value):
        self._numint.omega = value

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
        if mol is None: mol = self.mol
        if dm is None: dm = self.make_rdm1()
        if np.ndim(dm) == 2:
            dm = np.repeat(np.asarray(dm)[None] * .5, 2, axis=0)
        if self.grids.coords is None:
            self.grids.build()
        ni = self._numint
        t0 = time.perf_counter()
        n, exc, vxc = ni.nr_uks(mol, self.grids, self.xc, dm)
        self._log('nelec by numeric integration = %s; vxc %.4f s', n, time.perf_counter() - t0)
        omega, alpha, hyb = ni.rsh_and_hybrid_coeff(self.xc, spin=mol.spin)
        dma = np.asarray(dm)
        if hyb == 0 and (omega == 0 or alpha == 0):
            vk = None
            vj, _ = self.get_jk(mol, dm, hermi, with_k=False)
            vj = vj[0] + vj[1]
            vxc = vxc + vj
        else:
            # range-separated exact exchange, the reference's four branches (uks.py:84-105)
            if omega == 0:
                vj, vk = self.get_jk(mol, dm, hermi)
                vk = vk * hyb
            elif alpha == 0:               # short-range exchange only: the erfc-attenuated tensor
                vj = self.get_jk(mol, dm, hermi, with_k=False)[0]
                vk = self.get_jk(mol, dm, hermi, with_j=False, omega=-omega)[1] * hyb
            elif hyb == 0:                 # long-range exchange only
                vj = self.get_jk(mol, dm, hermi, with_k=False)[0]
                vk = self.get_jk(mol, dm, hermi, with_j=False, omega=omega)[1] * alpha
            else:                          # K = hyb K_full + (alpha - hyb) K_LR(omega)
                vj, vk = self.get_jk(mol, dm, hermi)
                vk = vk * hyb + self.get_jk(mol, dm, hermi, with_j=False, omega=omega)[1] * (alpha - hyb)
            vj = vj[0] + vj[1]
            vxc = vxc + vj - vk
            exc -= (np.einsum('ij,ji', dma[0], vk[0]) + np.einsum('ij,ji', dma[1], vk[1])).real * .5
        ecoul = np.einsum('ij,ji', dma[0] + dma[1], vj).real * .5
        return tag_array(vxc, ecoul=ecoul, exc=exc, vj=vj, vk=vk)

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        if dm is None: dm = self.make_rdm1()
        if h1e is None: h1e = self.get_hcore()
        if vhf is None or getattr(vhf, 'ecoul', None) is None:
            vhf = self.get_veff(self.mol, dm)
        dma = np.asarray(dm)
        e1 = np.einsum('ij,ji->', h1e, dma[0] + dma[1]).real
        e2 = vhf.ecoul.real + vhf.exc.real
        self.scf_summary.update(e1=e1, coul=vhf.ecoul.real, exc=vhf.exc.real)
        return e1 + e2, e2
