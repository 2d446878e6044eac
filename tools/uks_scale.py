"""Open-shell path at config-3 size: DF-UKS B3LYP SCF of the (H2O)_32 cc-pVTZ radical cation (charge +1, doublet), wall time
and energy; the same run with NumInt.sparse = False on the first Fock build checks the block-sparse nr_uks against the dense
pipeline at this size.
    python tools/uks_scale.py [--nwater 32 --basis cc-pvtz --xc b3lyp]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, dft
from pyscf_amd.data import clusters
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--xc', default='b3lyp')
a = ap.parse_args()
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis, charge=1, spin=1, verbose=4)
mf = dft.UKS(mol, xc=a.xc).density_fit()
mf.conv_tol = 1e-9
t0 = time.perf_counter()
e = mf.kernel()
wall = time.perf_counter() - t0
dm = mf.make_rdm1()
ni = mf._numint
n1, e1, v1 = ni.nr_uks(mol, mf.grids, a.xc, dm)
torch.cuda.synchronize()
t0 = time.perf_counter()
n1, e1, v1 = ni.nr_uks(mol, mf.grids, a.xc, dm)
torch.cuda.synchronize()
t_sparse = time.perf_counter() - t0
ref = dft.NumInt()
ref.sparse = False
t0 = time.perf_counter()
n0, e0, v0 = ref.nr_uks(mol, mf.grids, a.xc, dm)
torch.cuda.synchronize()
t_dense = time.perf_counter() - t0
print(json.dumps({'system': '(H2O)_%d+ %s UKS %s' % (a.nwater, a.basis, a.xc), 'nao': mol.nao, 'converged': bool(mf.converged),
                  'cycles': mf.cycles, 'e_tot': e, 'scf_wall_s': round(wall, 1), 'nr_uks_sparse_ms': round(t_sparse * 1e3, 1),
                  'nr_uks_dense_ms': round(t_dense * 1e3, 1), 'nelec': [float(x) for x in n1],
                  'sparse_vs_dense_max_abs_vmat': float(np.abs(v1 - v0).max()), 'sparse_vs_dense_exc': float(abs(e1 - e0)),
                  'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
