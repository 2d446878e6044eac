"""Where does the host API `with_df.get_jk(dm)` spend its time beyond the device-resident step?  (r05 diagnostic)
    python tools/host_api_probe.py --j2-policy overlap|fused|serial"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyscf_amd import gto, df, lib
from pyscf_amd.data import clusters
from pyscf_amd.df import df_jk
from pyscf_amd.scf import hf
ap = argparse.ArgumentParser()
ap.add_argument('--j2-policy', default='overlap')
ap.add_argument('--own-tag', type=int, default=0)
a = ap.parse_args()
mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
nao, nocc = mol.nao, mol.nelectron // 2
obj = df.DF(mol)
obj.j2_policy = a.j2_policy
obj.build()
dev = obj._cderi_dev.device
s1e = hf.int1e_gpu(mol, dev)[0]
x = np.random.RandomState(1).random_sample((nao, nao))
w, v = np.linalg.eigh(x.T.dot(s1e).dot(x))
c = x.dot(v / np.sqrt(w)).dot(v.T)
occ = np.zeros(nao); occ[:nocc] = 2
orbo = c[:, :nocc] * np.sqrt(2.0)
dm = orbo.dot(orbo.T)
tag = lib.tag_array(dm, mo_coeff=c, mo_occ=occ, **({'dm_from_orbitals': True} if a.own_tag else {}))
for _ in range(2):
    obj.get_jk(tag, hermi=1)
torch.cuda.synchronize()
import cProfile, pstats, gc
_g = {}


def _gc_cb(phase, info):
    if phase == 'start':
        _g['t'] = time.perf_counter()
    else:
        print('   gc gen %d: %.1f ms, collected %d' % (info['generation'], (time.perf_counter() - _g['t']) * 1e3, info['collected']), flush=True)


gc.callbacks.append(_gc_cb)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); obj.get_jk(tag, hermi=1); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('policy', a.j2_policy, 'own_tag', a.own_tag, 'host API ms per call', [round(t, 1) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    obj.get_jk(tag, hermi=1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(18)
