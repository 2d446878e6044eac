"""BASELINE config 3 end to end: converged total energies of (H2O)_32 cc-pVTZ against the CPU oracle's own SCF on its own
tensor (tests/golden/h2o32_ccpvtz_oracle.json, tools/gen_golden_fullsize.py).  Separate module so that the 184 GB of the
tensor fixture in test_gpu_fullsize.py are released first."""
import pytest

from tests.test_gpu_fullsize import _golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('xc', ['', 'b3lyp'])
def test_config3_converged_energy_vs_oracle_golden(xc):
    """BASELINE config 3 itself: (H2O)_32 cc-pVTZ DF-RHF and DF-RKS B3LYP converged total energies against the oracle's
    own SCF on its own 61 GB tensor (tests/golden/h2o32_ccpvtz_oracle.json), 1e-8 Eh."""
    import torch
    from pyscf_amd import gto, scf, dft
    from pyscf_amd.data import clusters
    g = _golden('h2o32_ccpvtz_oracle.json')
    key = 'e_rks_' + xc if xc else 'e_rhf'
    if key not in g:
        pytest.skip(key + ' not in the golden file')
    mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
    mf = (dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged
    assert abs(e - g[key]) < 1e-8, (e, g[key])
    mf.with_df.reset()
    del mf
    torch.cuda.empty_cache()
