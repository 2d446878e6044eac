"""GPU parity of mf.gen_response() (pyscf/scf/_response_functions.py): the Fock response to a first-order density equals
the derivative of get_veff along that density (central differences on the device), for every exchange branch."""
import numpy as np
import pytest

from tests.conftest import H2O

pytestmark = pytest.mark.gpu


def _perturbations(c_occ, seed, n=2):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        a = rng.standard_normal((c_occ.shape[1], c_occ.shape[1]))
        out.append(c_occ.dot(a + a.T).dot(c_occ.T))         # inside the occupied space: rho1 / rho0 stays bounded
    return np.array(out)


@pytest.mark.parametrize('xc', [None, 'lda,vwn', 'b3lyp', 'lda+0.5*SR_HF(0.3)', 'lda+0.4*LR_HF(1.0)',
                                'b88+0.2*HF+0.3*LR_HF(1.0),lyp'])
def test_rhf_rks_response_is_the_derivative_of_veff(xc):
    from pyscf_amd import gto, scf, dft
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = (scf.RHF(mol) if xc is None else dft.RKS(mol, xc=xc)).density_fit(auxbasis='weigend').run(conv_tol=1e-10)
    dm0 = mf.make_rdm1()
    co = mf.mo_coeff[:, mf.mo_occ > 0]
    d1 = _perturbations(co, 7)
    vind = mf.gen_response(hermi=1)
    v1 = vind(d1)
    assert v1.shape == d1.shape
    eps = 1e-4
    for i in range(len(d1)):
        fd = (np.asarray(mf.get_veff(mol, dm0 + eps * d1[i])) - np.asarray(mf.get_veff(mol, dm0 - eps * d1[i]))) / (2 * eps)
        assert np.abs(fd - v1[i]).max() < 3e-6 * max(1.0, np.abs(v1[i]).max()), (xc, i, np.abs(fd - v1[i]).max())
    # singlet TDDFT kernel = orbital-Hessian kernel; non-symmetric input (hermi = 0): K keeps the antisymmetric part
    vs = mf.gen_response(singlet=True, hermi=1)(d1[0])
    assert np.abs(vs - v1[0]).max() < 1e-10
    x = co.dot(np.random.default_rng(1).standard_normal((co.shape[1], mol.nao - co.shape[1]))).dot(
        mf.mo_coeff[:, mf.mo_occ == 0].T)
    full = mf.gen_response(hermi=0)(x)
    sym = mf.gen_response(hermi=1)((x + x.T) * .5)
    anti = mf.gen_response(hermi=2)((x - x.T) * .5)
    assert np.abs(full - (sym + anti)).max() < 1e-9 * max(1.0, np.abs(full).max())


@pytest.mark.parametrize('xc', ['lda,vwn', 'b3lyp'])
def test_triplet_and_uks_response(xc):
    """Triplet kernel of a closed-shell reference and the UKS response against derivatives of UKS.get_veff."""
    from pyscf_amd import gto, dft
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    mf = dft.RKS(mol, xc=xc).density_fit(auxbasis='weigend').run(conv_tol=1e-10)
    dm0 = mf.make_rdm1()
    co = mf.mo_coeff[:, mf.mo_occ > 0]
    d1 = _perturbations(co, 9, 1)[0]
    vt = mf.gen_response(singlet=False, hermi=1)(d1)
    mu = dft.UKS(mol, xc=xc).density_fit(auxbasis='weigend')
    mu.grids = mf.grids
    eps = 1e-4
    half = dm0 * .5

    def veff_u(da, db):
        return np.asarray(mu.get_veff(mol, np.array((da, db))))
    fd = (veff_u(half + eps * d1 * .5, half - eps * d1 * .5) - veff_u(half - eps * d1 * .5, half + eps * d1 * .5)) / (2 * eps)
    assert np.abs(fd[0] - vt).max() < 3e-6 * max(1.0, np.abs(vt).max()), np.abs(fd[0] - vt).max()
    assert np.abs(fd[1] + vt).max() < 3e-6 * max(1.0, np.abs(vt).max())
    # open-shell cation
    molc = gto.M(atom=H2O, basis='cc-pvdz', charge=1, spin=1)
    mfc = dft.UKS(molc, xc=xc).density_fit(auxbasis='weigend').run(conv_tol=1e-10)
    da, db = mfc.make_rdm1()
    ca = mfc.mo_coeff[0][:, mfc.mo_occ[0] > 0]
    cb = mfc.mo_coeff[1][:, mfc.mo_occ[1] > 0]
    d1a, d1b = _perturbations(ca, 3, 1)[0], _perturbations(cb, 4, 1)[0]
    v1 = mfc.gen_response(hermi=1)(np.array((d1a, d1b)))
    fd = (np.asarray(mfc.get_veff(molc, np.array((da + eps * d1a, db + eps * d1b)))) -
          np.asarray(mfc.get_veff(molc, np.array((da - eps * d1a, db - eps * d1b))))) / (2 * eps)
    assert np.abs(fd - v1).max() < 3e-6 * max(1.0, np.abs(v1).max()), np.abs(fd - v1).max()


def test_lowrank_exchange_equals_general_branch():
    """K of factorised trial densities D = L R^T (+ h.c.) through two MO-branch half transforms + one X^T Y product
    (df_jk._vk_lowrank) against the general-DM branch (pyscf/df/df_jk.py:382-407) on the same matrices; shared left
    factors, several densities, symmetric and plain forms, the square-image and the packed-operand half transforms."""
    import numpy as np
    from pyscf_amd import gto, df, lib
    from pyscf_amd.data import clusters
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')
    nao, nocc = mol.nao, mol.nelectron // 2
    rng = np.random.default_rng(9)
    co = np.linalg.qr(rng.standard_normal((nao, nocc)))[0]
    co2 = np.linalg.qr(rng.standard_normal((nao, nocc)))[0]
    rs = rng.standard_normal((3, nao, nocc)) * 0.1
    for square in ('auto', False):
        obj = df.DF(mol)
        obj.k_square = square
        obj.build()
        for sym in (False, True):
            lefts = [co, co, co2]
            dms = np.array([l.dot(r.T) for l, r in zip(lefts, rs)])
            if sym:
                dms = dms + dms.transpose(0, 2, 1)
            vj0, vk0 = obj.get_jk(dms, hermi=0)
            vj1, vk1 = obj.get_jk(lib.tag_array(dms, lowrank=(lefts, list(rs), sym)), hermi=0)
            assert np.abs(vj1 - vj0).max() < 1e-11 and np.abs(vk1 - vk0).max() < 1e-11 * max(1.0, np.abs(vk0).max())
            vk2 = obj.get_jk(lib.tag_array(dms, lowrank=(lefts, list(rs), sym)), hermi=0, with_j=False)[1]
            assert np.abs(vk2 - vk0).max() < 1e-11 * max(1.0, np.abs(vk0).max())
