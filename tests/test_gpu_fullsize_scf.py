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
    if key + '_unseeded' in g:                     # the oracle's independent run (its own start density, no product data)
        assert 'no product data' in g[key + '_unseeded_note']
        assert abs(e - g[key + '_unseeded']) < 1e-8, (e, g[key + '_unseeded'])
    mf.with_df.reset()
    del mf
    torch.cuda.empty_cache()


def test_config3_rhf_energy_vs_independent_oracle_scf():
    """The oracle's INDEPENDENT SCF at BASELINE config 3: converged by oracle/ref.rhf_kernel from a start
    density the oracle made alone (superposition of its own monomer densities; tools/gen_golden_fullsize.py --unseeded, log in
    tests/golden/h2o32_oracle_rhf_unseeded.log) - no product data on that path, unlike `e_rhf` whose run started from a dumped
    product density.  The product starts from its MINAO guess: two different starts, two different codes, one energy to 1e-8 Eh."""
    import torch
    from pyscf_amd import gto, scf
    from pyscf_amd.data import clusters
    g = _golden('h2o32_ccpvtz_oracle.json')
    if 'e_rhf_unseeded' not in g:
        pytest.skip('e_rhf_unseeded not in the golden file')
    assert 'no product data' in g['e_rhf_unseeded_note']
    assert abs(g['e_rhf_unseeded'] - g['e_rhf']) < 1e-8              # the oracle's two runs agree with each other
    mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
    mf = scf.RHF(mol).density_fit()
    mf.conv_tol = 1e-10
    e = mf.kernel()
    assert mf.converged and abs(e - g['e_rhf_unseeded']) < 1e-8, (e, g['e_rhf_unseeded'])
    homo, lumo = mf.mo_energy[mol.nelectron // 2 - 1], mf.mo_energy[mol.nelectron // 2]
    assert abs(homo - g['homo_lumo_unseeded'][0]) < 1e-6 and abs(lumo - g['homo_lumo_unseeded'][1]) < 1e-6
    mf.with_df.reset()
    del mf
    torch.cuda.empty_cache()
