"""Host-side profile of df_jk.get_jk_device on one rank's shard (launch overhead at small shards).
    python tools/prof_jk_host.py --world 8"""
import argparse, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, df
from pyscf_amd.data import clusters
from pyscf_amd.df import df_jk
ap = argparse.ArgumentParser()
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
a = ap.parse_args()
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
nao, nocc = mol.nao, mol.nelectron // 2
obj = df.DF(mol)
obj._shard_override = (a.world // 2, a.world)
obj.build()
rng = np.random.default_rng(1)
c = np.linalg.qr(rng.standard_normal((nao, nocc)))[0] * np.sqrt(2.0)
dev = obj._cderi_dev.device
dm = torch.from_numpy(c.dot(c.T)[None]).to(dev)
orb = [df_jk.pad_orbitals(c, dev)]
for _ in range(3):
    df_jk.get_jk_device(obj, dm, orb)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    df_jk.get_jk_device(obj, dm, orb)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
obj.kernel_timer = df_jk.KernelTimer()
for _ in range(5):
    df_jk.get_jk_device(obj, dm, orb)
s = obj.kernel_timer.summary()
obj.kernel_timer = None
print('rows %d  wall %.3f ms  kernels (serial sum) %.3f ms: %s' % (obj._cderi_dev.shape[0], wall, sum(t for t, _ in s.values()) / 5,
      {k: round(t / 5, 3) for k, (t, _) in s.items()}))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    df_jk.get_jk_device(obj, dm, orb)
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(18)
print(out.getvalue()[:3500])
