"""Second-order SCF (mf.newton(), pyscf/soscf/newton_ah.py) on the device response path: same energies as the DIIS driver
and as the reference's golden values, quadratic convergence (few Fock builds), canonical orbitals afterwards."""
import numpy as np
import pytest

from tests.conftest import H2O

pytestmark = pytest.mark.gpu


def test_newton_rhf_reaches_the_golden_energy():
    """DF-RHF H2O / cc-pVDZ: -76.025936299702536 (pyscf/df/test/test_df_jk.py:44-47)."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = scf.RHF(mol).density_fit(auxbasis='weigend')
    mf.conv_tol = 1e-11
    nt = mf.newton()
    e = nt.kernel()
    assert nt.converged and abs(e - -76.025936299702536) < 1e-9, e
    assert nt.cycles <= 9, nt.cycles
    # the wrapped object now holds canonical orbitals: its analytic gradient equals the DIIS one
    g = mf.nuc_grad_method().kernel()
    ref_mf = scf.RHF(mol).density_fit(auxbasis='weigend').run(conv_tol=1e-11)
    assert np.abs(np.sort(mf.mo_energy) - np.sort(ref_mf.mo_energy)).max() < 1e-6
    assert np.abs(g - ref_mf.nuc_grad_method().kernel()).max() < 1e-6


@pytest.mark.parametrize('xc', ['lda,vwn', 'b3lyp', 'lda+0.5*SR_HF(0.3)'])
def test_newton_rks_equals_diis(xc):
    from pyscf_amd import gto, dft
    atoms = [('O', (0., 0., 0.)), ('H', (0., -0.757, 0.587)), ('H', (0.3, 1.1, 0.7))]       # one stretched bond
    mol = gto.M(atom=atoms, basis='cc-pvdz')
    ref_mf = dft.RKS(mol, xc=xc).density_fit().run(conv_tol=1e-11)
    assert ref_mf.converged
    mf = dft.RKS(mol, xc=xc).density_fit()
    mf.conv_tol = 1e-11
    mf.grids = ref_mf.grids
    nt = mf.newton()
    e = nt.kernel()
    assert nt.converged and abs(e - ref_mf.e_tot) < 2e-9, (e, ref_mf.e_tot)
    assert nt.cycles <= 10 and nt.hessian_products < 80, (nt.cycles, nt.hessian_products)
    # restart from converged orbitals: one Fock build, no step
    nt2 = mf.newton()
    e2 = nt2.kernel(mf.mo_coeff, mf.mo_occ)
    assert nt2.cycles == 1 and abs(e2 - e) < 1e-10
