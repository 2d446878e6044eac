"""Analytic nuclear gradients of density-fitted UKS on the MI355X path (pyscf/grad/uks.py get_veff :37-98 /
get_vxc :100-190 with grid_response=False, on top of pyscf/df/grad/uks.py): the J / hyb*K part is
``grad.rhf.grad_elec_df`` with the two occupied blocks, the XC part ``NumInt.nr_uks_grad``."""
import numpy as np

from . import rhf as rhf_grad


class Gradients(rhf_grad.Gradients):
    grid_response = False

    def grad_elec(self):
        mf = self.base
        if getattr(mf, 'with_df', None) is None:
            raise NotImplementedError('gradients are implemented for density-fitted SCF objects')
        ni = mf._numint
        omega, alpha, hyb = ni.rsh_and_hybrid_coeff(mf.xc, spin=self.mol.spin)
        dm, blocks, dme = self._densities()
        kfull, extra = rhf_grad.rsh_exchange_terms(mf.with_df, omega, alpha, hyb)
        de = rhf_grad.grad_elec_df(self.mol, mf.with_df, dm, blocks, dme, kfull, self.auxbasis_response,
                                   exchange_terms=extra)
        dms = [c.dot(c.T) for c, _w in blocks]
        return de + ni.nr_uks_grad(self.mol, mf.grids, mf.xc, dms, self.grid_response)
