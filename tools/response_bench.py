"""Cost of the response path at scale: one mf.gen_response() product (and one TDA Davidson-sized batch) on (H2O)_n.
    python tools/response_bench.py [--nwater 32 --basis cc-pvtz --xc b3lyp --nvec 4]
No SCF is run: orbitals come from one diagonalisation of the Fock matrix of the minao guess (timing is independent of
convergence)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, scf, dft
from pyscf_amd.data import clusters

ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--xc', default='b3lyp')
ap.add_argument('--nvec', type=int, default=4)
ap.add_argument('--skip-general', action='store_true')
ap.add_argument('--tune', default='', help='comma list key=value for PAMD_set_tuning')
a = ap.parse_args()
from pyscf_amd import lib as _L
for kv in filter(None, a.tune.split(',')):
    k, v = kv.split('=')
    _L.check(_L.load_library().PAMD_set_tuning(k.encode(), int(v)))
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
mf = (dft.RKS(mol, xc=a.xc) if a.xc else scf.RHF(mol)).density_fit()
t0 = time.perf_counter()
mf.with_df.build()
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
dm0 = mf.get_init_guess(mol, 'minao')
h1e, s1e = mf.get_hcore(), mf.get_ovlp()
vhf = mf.get_veff(mol, dm0)
e, c = mf.eig(h1e + np.asarray(vhf), s1e, mf.check_linear_dependency(s1e))
occ = mf.get_occ(e, c)
co, cv = c[:, occ > 0], c[:, occ == 0]
rng = np.random.default_rng(1)
out = {'xc': a.xc or 'hf', 'nwater': a.nwater, 'nao': mol.nao, 'naux': mf.with_df.get_naoaux(), 'nocc': co.shape[1],
       'tensor_build_s': round(t_build, 2)}


def timed(fn, *args):
    fn(*args)                                   # warm-up (grids, square image, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(*args)
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0
from pyscf_amd import lib
for hermi, label in ((1, 'orbital_hessian_product_s'), (0, 'tddft_product_s')):
    vind = mf.gen_response(c, occ, hermi=hermi)
    x = rng.standard_normal((co.shape[1], cv.shape[1])) * 1e-2
    r = 2 * cv.dot(x.T)
    d1 = co.dot(r.T)
    if hermi == 1:
        d1 = d1 + d1.T
    # as the solvers call it: the trial density carries its factors (low-rank exchange, orbital-product densities)
    v, t = timed(vind, lib.tag_array(d1, lowrank=([co], [r], hermi == 1)))
    out[label] = round(t, 3)
    if not a.skip_general:
        v0, t0_ = timed(vind, d1)                    # untagged: general-DM exchange + eigen-factorised density
        out[label.replace('_s', '_untagged_s')] = round(t0_, 3)
        out[label.replace('_s', '_tag_vs_untagged_maxdiff')] = float(abs(v - v0).max())
    if hermi == 0:
        xs = rng.standard_normal((a.nvec, co.shape[1], cv.shape[1])) * 1e-2
        rs = 2 * np.matmul(cv, xs.transpose(0, 2, 1))
        dms = np.matmul(co, rs.transpose(0, 2, 1))
        v, t = timed(vind, lib.tag_array(dms, lowrank=([co] * a.nvec, list(rs), False)))
        out['tddft_batch_of_%d_s' % a.nvec] = round(t, 3)
out['peak_hbm_gb'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
print(json.dumps(out))
