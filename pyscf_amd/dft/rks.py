"""Restricted Kohn-Sham driver on top of the SCF driver.

Mirror of ``pyscf/dft/rks.py``: ``get_veff`` (:37-142), ``energy_elec`` (:228-258),
``KohnShamDFT`` defaults (:273-524: xc='LDA,VWN', grids level 3, small_rho_cutoff handled as 0).
J (and K for hybrids) go through ``mf.with_df.get_jk`` exactly like the DF-RHF path; the XC
matrix comes from ``mf._numint.nr_rks`` (pyscf_amd/dft/numint.py)."""
import time

import numpy as np

from ..lib import tag_array
from ..scf import hf
from . import gen_grid, numint


def get_veff(ks, mol=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
    if mol is None: mol = ks.mol
    if dm is None: dm = ks.make_rdm1()
    if ks.grids.coords is None:
        t0 = time.perf_counter()
        ks.grids.build()
        ks._log('setting up grids: %d points, %.2f s', ks.grids.size, time.perf_counter() - t0)
    ni = ks._numint
    t0 = time.perf_counter()
    n, exc, vxc = ni.nr_rks(mol, ks.grids, ks.xc, dm)
    ks._log('nelec by numeric integration = %s; vxc %.4f s', n, time.perf_counter() - t0)
    omega, alpha, hyb = ni.rsh_and_hybrid_coeff(ks.xc, spin=mol.spin)
    t0 = time.perf_counter()
    if hyb == 0 and (omega == 0 or alpha == 0):
        vk = None
        vj, _ = ks.get_jk(mol, dm, hermi, with_k=False)
        vxc = vxc + vj
    else:
        # range-separated exact exchange, the reference's four branches (rks.py:110-127)
        if omega == 0:
            vj, vk = ks.get_jk(mol, dm, hermi)
            vk = vk * hyb
        elif alpha == 0:                    # short-range exchange only: the erfc-attenuated tensor
            vj = ks.get_jk(mol, dm, hermi, with_k=False)[0]
            vk = ks.get_jk(mol, dm, hermi, with_j=False, omega=-omega)[1] * hyb
        elif hyb == 0:                      # long-range exchange only
            vj = ks.get_jk(mol, dm, hermi, with_k=False)[0]
            vk = ks.get_jk(mol, dm, hermi, with_j=False, omega=omega)[1] * alpha
        else:                               # K = hyb K_full + (alpha - hyb) K_LR(omega)
            vj, vk = ks.get_jk(mol, dm, hermi)
            vk = vk * hyb + ks.get_jk(mol, dm, hermi, with_j=False, omega=omega)[1] * (alpha - hyb)
        vxc = vxc + vj - vk * .5
        exc -= np.einsum('ij,ji', dm, vk).real * .5 * .5
    ks._log('df vj and vk: %.4f s', time.perf_counter() - t0)
    ecoul = np.einsum('ij,ji', dm, vj).real * .5
    return tag_array(vxc, ecoul=ecoul, exc=exc, vj=vj, vk=vk)


def energy_elec(ks, dm=None, h1e=None, vhf=None):
    if dm is None: dm = ks.make_rdm1()
    if h1e is None: h1e = ks.get_hcore()
    if vhf is None or getattr(vhf, 'ecoul', None) is None:
        vhf = ks.get_veff(ks.mol, dm)
    e1 = np.einsum('ij,ji->', h1e, dm).real
    ecoul = vhf.ecoul.real
    exc = vhf.exc.real
    e2 = ecoul + exc
    ks.scf_summary.update(e1=e1, coul=ecoul, exc=exc)
    return e1 + e2, e2


class RKS(hf.RHF):
    def __init__(self, mol, xc='LDA,VWN'):
        hf.RHF.__init__(self, mol)
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
        return get_veff(self, mol, dm, dm_last, vhf_last, hermi)

    def energy_elec(self, dm=None, h1e=None, vhf=None):
        return energy_elec(self, dm, h1e, vhf)

    def density_fit(self, auxbasis=None, with_df=None, only_dfj=False, devices=None):
        # (pure functionals fit J only: the reference picks a J-fit set, df/addons.py:326-330, 354-357 - DF(mol, None) follows the
        # same rule); `devices`: scf/hf.py::SCF.density_fit - J/K and the XC tiles over a device list in this process
        return hf.SCF.density_fit(self, auxbasis, with_df, only_dfj, devices)
