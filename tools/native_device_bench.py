"""Both orchestrations on the SAME device-resident inputs: config 3, one J/K build per step,
  (a) pyscf_amd.df.DF + df_jk.get_jk_device (torch layer: the bench line's path),
  (b) NativeDF.get_jk_device (C handle, PAMD_df_get_jk with device pointers, flags bit 3).
    python tools/native_device_bench.py [--steps 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, df
from pyscf_amd.data import clusters
from pyscf_amd.df import df_jk
from pyscf_amd.df.native import NativeDF
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--which', default='handle,torch')
a = ap.parse_args()
os.environ.setdefault('PAMD_DF_J2_TUNE', 'eager')
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
nao, nocc = mol.nao, mol.nelectron // 2
dev = torch.device('cuda', 0)
c = np.linalg.qr(np.random.RandomState(1).rand(nao, nao))[0][:, :nocc] * np.sqrt(2.0)
orbo = torch.from_numpy(np.ascontiguousarray(c)).to(dev)
dm = orbo @ orbo.T
out = {'nao': nao, 'nocc': nocc, 'steps': a.steps}
ref = None
for which in a.which.split(','):
    if which == 'handle':
        obj = NativeDF(mol).build()
        run = lambda: obj.get_jk_device(dm, orbo)
    else:
        obj = df.DF(mol)
        obj.j2_tune = 'eager'
        obj.build()
        orb = [df_jk.pad_orbitals(c, dev)]
        def run():
            vjt, vk = df_jk.get_jk_device(obj, dm[None], orb, True, True, dm_from_orbitals=True)
            return vjt[0], vk[0]
    for _ in range(3):
        vj, vk = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        vj, vk = run()
    torch.cuda.synchronize()
    out[which + '_ms'] = round((time.perf_counter() - t0) / a.steps * 1e3, 2)
    fp = float(vk.sum())
    if ref is None:
        ref = fp
    out[which + '_fp_vk'] = fp
    del obj, run
    torch.cuda.empty_cache()
print(json.dumps(out))
