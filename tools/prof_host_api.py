"""Where does the host-API J/K call (`with_df.get_jk(dm)` with numpy in / out, the reference's 'df vj and vk' timer,
pyscf/df/df_jk.py:412) spend its time beyond the device-resident build?  cProfile of a few calls + fused / unfused J check.
    python tools/prof_host_api.py [--nwater 32]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, df, lib
from pyscf_amd.data import clusters
from pyscf_amd.scf import hf
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
a = ap.parse_args()
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
nao, nocc = mol.nao, mol.nelectron // 2
obj = df.DF(mol).build()
s1e = hf.int1e_gpu(mol)[0]
rng = np.random.RandomState(1)
x = rng.random_sample((nao, nao))
w, v = np.linalg.eigh(x.T.dot(s1e).dot(x))
c = x.dot(v / np.sqrt(w)).dot(v.T)
occ = np.zeros(nao); occ[:nocc] = 2
orbo = c[:, :nocc] * np.sqrt(2.0)
dm = orbo.dot(orbo.T)
for name, tag in (('foreign tag', lib.tag_array(dm, mo_coeff=c, mo_occ=occ)),
                  ('promised tag', lib.tag_array(dm, mo_coeff=c, mo_occ=occ, dm_from_orbitals=True))):
    obj.get_jk(tag, hermi=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        obj.get_jk(tag, hermi=1)
    torch.cuda.synchronize()
    print('%s: %.1f ms per call, fused J = %s' % (name, (time.perf_counter() - t0) / 3 * 1e3, getattr(obj, '_last_fused', None)), flush=True)
pr = cProfile.Profile()
tag = lib.tag_array(dm, mo_coeff=c, mo_occ=occ)
pr.enable()
for _ in range(3):
    obj.get_jk(tag, hermi=1)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
