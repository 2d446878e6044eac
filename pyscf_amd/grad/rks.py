"""Analytic nuclear gradients of density-fitted RKS on the MI355X path.

Mirror of ``pyscf/grad/rks.py`` (``get_veff`` :37-116, ``get_vxc`` :119-195 with ``grid_response=False``, the default)
on top of ``pyscf/df/grad/rks.py`` (J from the fitted density, ``hyb`` x K for hybrids).  The Coulomb/exchange part is
``grad.rhf.grad_elec_df`` with the exchange scaled by the hybrid coefficient; the XC part is
``NumInt.nr_rks_grad`` (AO Hessians from ``PAMD_eval_ao(deriv=2)``, reduction ``PAMD_xc_grad``)."""
import numpy as np

from . import rhf as rhf_grad


class Gradients(rhf_grad.Gradients):
    grid_response = False

    def grad_elec(self):
        mf = self.base
        if getattr(mf, 'with_df', None) is None:
            raise NotImplementedError('gradients are implemented for density-fitted SCF objects')
        mo_occ = np.asarray(mf.mo_occ)
        if mo_occ.ndim != 1:
            raise NotImplementedError('UKS gradients')
        ni = mf._numint
        omega, alpha, hyb = ni.rsh_and_hybrid_coeff(mf.xc, spin=self.mol.spin)
        dm, blocks, dme = self._densities()
        kfull, extra = rhf_grad.rsh_exchange_terms(mf.with_df, omega, alpha, hyb)
        de = rhf_grad.grad_elec_df(self.mol, mf.with_df, dm, blocks, dme, kfull, self.auxbasis_response,
                                   exchange_terms=extra)
        return de + ni.nr_rks_grad(self.mol, mf.grids, mf.xc, dm, self.grid_response)
