"""Converged SCF density of (H2O)_n through the product path, saved as .npy: a START density for the oracle's own SCF in
tools/gen_golden_fullsize.py (--dm0; the oracle's converged energy does not depend on where it starts).
    python tools/dump_scf_dm.py --nwater 32 --xc b3lyp --out gpurun_out/dm_h2o32_b3lyp.npy"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_amd import gto, scf, dft
from pyscf_amd.data import clusters
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--xc', default='')
ap.add_argument('--out', required=True)
a = ap.parse_args()
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis, verbose=4)
mf = (dft.RKS(mol, xc=a.xc) if a.xc else scf.RHF(mol)).density_fit()
mf.conv_tol = 1e-10
t0 = time.perf_counter()
e = mf.kernel()
print('converged=%s cycles=%d E=%.12f wall=%.1f s' % (mf.converged, mf.cycles, e, time.perf_counter() - t0), flush=True)
np.save(a.out, np.asarray(mf.make_rdm1()))
