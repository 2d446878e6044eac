"""Build and contract ONE aux-row shard of a large configuration on a single GPU (emulates rank r of N
without a process group): checks memory footprint, 64-bit indexing and timings at BASELINE config-5 scale.
    python tools/shard_probe.py --nwater 128 --basis cc-pvdz --world 8 --rank 3
    python tools/shard_probe.py --molecule taxol --basis def2-tzvp --world 8 --rank 3        (config 4)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, df
from pyscf_amd.data import clusters
from pyscf_amd.df import df_jk
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=128)
ap.add_argument('--basis', default='cc-pvdz')
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--rank', type=int, default=3)
ap.add_argument('--molecule', default='water', choices=['water', 'taxol'])
ap.add_argument('--repeat', type=int, default=3)
ap.add_argument('--no-image', action='store_true', help='square rows as the only copy whenever 2x fits')
a = ap.parse_args()
mol = gto.M(atom=clusters.taxol() if a.molecule == 'taxol' else clusters.water_cluster(a.nwater), basis=a.basis)
nao, nocc = mol.nao, mol.nelectron // 2
obj = df.DF(mol)
obj._shard_override = (a.rank, a.world)
t0 = time.perf_counter()
if '--no-image' in sys.argv:
    obj.prefer_image = False
obj.build()
torch.cuda.synchronize()
tb = time.perf_counter() - t0
shard_rows, npair = obj.tensor_shape()             # (never `_cderi_dev` here: in the square layout that property packs a copy)
naux = obj.get_naoaux()
hbm_after_build = torch.cuda.memory_allocated() * 1e-9
rng = np.random.default_rng(1)
c = np.linalg.qr(rng.standard_normal((nao, nocc)))[0] * np.sqrt(2.0)
dev = obj.tensor_device()
dm = torch.from_numpy(c.dot(c.T)[None]).to(dev)
orb = [df_jk.pad_orbitals(c, dev)]
vj, vk = df_jk.get_jk_device(obj, dm, orb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.repeat):
    vj, vk = df_jk.get_jk_device(obj, dm, orb)
torch.cuda.synchronize()
tjk = (time.perf_counter() - t0) / a.repeat
# spot check against a dense fp64 reference on 8 rows of the shard
sub = obj.packed_rows(0, 8)
idx = torch.tril_indices(nao, nao, device=dev)
full = torch.zeros((8, nao, nao), dtype=torch.float64, device=dev)
full[:, idx[0], idx[1]] = sub
full = full + full.transpose(1, 2) - torch.diag_embed(torch.diagonal(full, dim1=1, dim2=2))
o2 = df.DF(mol); o2._cderi_dev = sub
vj2, vk2 = df_jk.get_jk_device(o2, dm, orb)
cdv = torch.from_numpy(c).to(dev)
xx = torch.matmul(full, cdv)
vk_ref = torch.einsum('Lpi,Lqi->pq', xx, xx)
rho = torch.einsum('Lpq,pq->L', full, dm[0])
vj_ref = torch.einsum('L,Lpq->pq', rho, full)
vjf = torch.zeros((nao, nao), dtype=torch.float64, device=dev); vjf[idx[0], idx[1]] = vj2[0]
print(json.dumps({'molecule': a.molecule if a.molecule != 'water' else '(H2O)_%d' % a.nwater, 'basis': a.basis, 'rank': a.rank,
                  'world': a.world, 'square_image': getattr(obj, '_cderi_sq', None) is not None, 'nao': nao, 'naux': naux, 'nocc': nocc, 'shard_rows': int(shard_rows),
                  'shard_GB': round(shard_rows * npair * 8e-9, 1), 'tensor_layout': obj._layout, 'hbm_after_build_GB': round(hbm_after_build, 1),
                  'hbm_after_jk_GB': round(torch.cuda.memory_allocated() * 1e-9, 1), 'hbm_free_after_jk_GB': round(torch.cuda.mem_get_info()[0] * 1e-9, 1), 'build_s': round(tb, 1), 'jk_ms_this_shard': round(tjk * 1e3, 1),
                  'mem_peak_GB': round(torch.cuda.max_memory_allocated() * 1e-9, 1),
                  'err_vk': float((vk2[0] - vk_ref).abs().max() / vk_ref.abs().max()),
                  'err_vj': float((vjf.tril() - vj_ref.tril()).abs().max() / vj_ref.abs().max()),
                  'finite': bool(torch.isfinite(vk).all() and torch.isfinite(vj).all())}))
