"""GPU parity: analytic DF-RHF / DF-UHF nuclear gradients (generate-and-contract kernels through the C ABI)
against the reference's known answers (pyscf/df/test/test_df_grad.py) and the finite-difference oracle."""
import numpy as np
import pytest

from oracle import ref, ref_grad

pytestmark = pytest.mark.gpu

BOHR = 0.52917721092
H2O = [('O', (0., 0., 0.)), ('H', (0., -0.757, 0.587)), ('H', (0., 0.757, 0.587))]


def _bohr(atoms):
    return [(s, tuple(np.array(r) / BOHR)) for s, r in atoms]


def test_df_rhf_gradient_goldens_and_fd():
    """test_df_grad.py:57-65: H2O 6-31G, aux cc-pvdz-jkfit; lib.fp(g) with and without the aux response."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='6-31g')
    mf = scf.RHF(mol).density_fit(auxbasis='ccpvdz-jkfit').run(conv_tol=1e-12)
    g0 = mf.Gradients().set(auxbasis_response=False).kernel()
    assert abs(ref.fp(g0) - 0.005466630382488041) < 2e-7
    g = mf.nuc_grad_method().kernel()
    assert abs(ref.fp(g) - 0.005516638190173352) < 2e-7
    assert abs(g.sum(axis=0)).max() < 1e-10                      # translational invariance
    assert abs(g[1, 1] + g[2, 1]) < 1e-10 and abs(g[1, 2] - g[2, 2]) < 1e-10
    gfd = ref_grad.fd_gradient(_bohr(H2O), '6-31g', 'ccpvdz-jkfit')
    assert np.abs(g - gfd).max() < 2e-7


def test_df_uhf_gradient_goldens_and_fd():
    """test_df_grad.py:95-110: triplet H2O 6-31G (default aux cc-pvdz-jkfit), UHF."""
    from pyscf_amd import gto, scf
    mol = gto.M(atom=H2O, basis='631g', spin=2)
    mf = scf.UHF(mol).density_fit().run(conv_tol=1e-12)
    g0 = mf.Gradients().set(auxbasis_response=False).kernel()
    assert abs(ref.fp(g0) - -0.19670644982746546) < 5e-7
    g = mf.Gradients().kernel()
    assert abs(ref.fp(g) - -0.19660674423263175) < 5e-7
    # the oracle's displaced SCFs start from this state's orbitals (its core guess lands on another triplet)
    gfd = ref_grad.fd_gradient(_bohr(H2O), '631g', None, spin=2, components=[(0, 2), (1, 1)], mo0=mf.mo_coeff)
    assert abs(g[0, 2] - gfd[0, 2]) < 1e-6 and abs(g[1, 1] - gfd[1, 1]) < 1e-6


@pytest.mark.parametrize('basis,aux', [('cc-pvdz', None), ('cc-pvtz', None), ('def2-svp', 'def2-universal-jkfit')])
def test_df_rhf_gradient_higher_l_vs_fd(basis, aux):
    """d and f AO shells (derivative classes up to l+1 = 4 in the recurrences), aux shells up to g;
    a geometry without symmetry so that every Cartesian component is exercised."""
    from pyscf_amd import gto, scf
    atoms = [('O', (0.03, -0.02, 0.01)), ('H', (0.1, -0.757, 0.587)), ('H', (-0.2, 0.8, 0.5))]
    mol = gto.M(atom=atoms, basis=basis)
    mf = scf.RHF(mol).density_fit(auxbasis=aux).run(conv_tol=1e-12)
    g = mf.nuc_grad_method().kernel()
    assert abs(g.sum(axis=0)).max() < 1e-9
    comps = [(0, 0), (1, 1), (2, 2)]
    gfd = ref_grad.fd_gradient(_bohr(atoms), basis, aux, components=comps)
    for a, x in comps:
        assert abs(g[a, x] - gfd[a, x]) < 5e-7, (a, x, g[a, x], gfd[a, x])
