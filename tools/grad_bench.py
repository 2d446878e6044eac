"""DF-RHF analytic gradient at scale: (H2O)_n cc-pVTZ, with a product-side finite-difference spot check.
    python tools/grad_bench.py [--nwater 8 --basis cc-pvtz --fd]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, scf, dft
from pyscf_amd.data import clusters

ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=8)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--fd', action='store_true')
ap.add_argument('--xc', default='')
ap.add_argument('--grid-response', action='store_true')
a = ap.parse_args()
atoms = clusters.water_cluster(a.nwater)
mol = gto.M(atom=atoms, basis=a.basis)
t0 = time.perf_counter()
mf = (dft.RKS(mol, xc=a.xc) if a.xc else scf.RHF(mol)).density_fit().run(conv_tol=1e-11)
t_scf = time.perf_counter() - t0
t0 = time.perf_counter()
gm = mf.nuc_grad_method()
if a.grid_response: gm.grid_response = True
g = gm.kernel()
torch.cuda.synchronize()
t_grad = time.perf_counter() - t0
out = {'xc': a.xc or 'hf', 'grid_response': a.grid_response, 'nwater': a.nwater, 'nao': mol.nao, 'naux': mf.with_df.get_naoaux(), 'e_tot': mf.e_tot, 'scf_s': round(t_scf, 2),
       'cycles': mf.cycles, 'grad_s': round(t_grad, 2), 'sum_over_atoms': np.abs(g.sum(0)).max(),
       'max_abs_grad': float(np.abs(g).max()), 'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2**30, 1)}
if a.fd:
    h = 2e-3
    r = mol.atom_coords()
    es = []
    for d in (h, -h):
        rr = r.copy(); rr[0, 2] += d
        m2 = gto.M(atom=[(mol.atom_symbol(i), tuple(rr[i])) for i in range(mol.natm)], basis=a.basis, unit='Bohr')
        es.append(scf.RHF(m2).density_fit().run(conv_tol=1e-11).e_tot)
    out['fd_g02'] = (es[0] - es[1]) / (2 * h)
    out['analytic_g02'] = float(g[0, 2])
print(json.dumps(out))
