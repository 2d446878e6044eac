"""End-to-end DF-SCF on a water cluster through the reference-style API.
    python tools/run_scf.py --nwater 32 --basis cc-pvtz --xc b3lyp     (xc '' -> RHF)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyscf_amd import gto, scf, dft
from pyscf_amd.data import clusters
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--xc', default='b3lyp')
ap.add_argument('--conv-tol', type=float, default=1e-9)
a = ap.parse_args()
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis, verbose=4)
mf = (dft.RKS(mol, xc=a.xc) if a.xc else scf.RHF(mol)).density_fit()
mf.conv_tol = a.conv_tol
t0 = time.perf_counter()
e = mf.kernel()
print('converged=%s cycles=%d E=%.10f wall=%.1f s (nao=%d naux=%d)' %
      (mf.converged, mf.cycles, e, time.perf_counter() - t0, mol.nao, mf.with_df.get_naoaux()), flush=True)
