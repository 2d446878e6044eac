"""Static tile sparsity of the DF tensor (VERDICT r05 item 5 / Weak 13).

The AO-pair pattern of B[L, pq] = sum_Q L^-1[L, Q] (Q|pq) is L-independent: a pair pq whose overlap distribution is negligible is
negligible for every aux row.  The half transform (e2_sq2 / e2_pk) walks, per 128-column tile of p, the 16-row k-tiles of q; a
k-tile that is below the threshold for EVERY L could be dropped from a static per-column-tile list at zero run-time cost
(reference semantics being preserved: pyscf/df/df_jk.py:339-381 contracts every element).  This tool measures how many such
tiles there are:  M[p, q] = max_L |B[L, pq]| over this rank's rows, then the live fraction of (16 q x 128 p) tiles (the kernels'
unit), of (16 x 16) tiles and of single pairs, at 1e-13 / 1e-12 / 1e-10.

    python tools/tile_sparsity.py --nwater 32 --basis cc-pvtz                       (config 3)
    python tools/tile_sparsity.py --molecule taxol --basis def2-tzvp                (config 4)
    python tools/tile_sparsity.py --nwater 128 --basis cc-pvdz --world 8 --rank 3   (one config-5 shard)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, df
from pyscf_amd.data import clusters

ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--molecule', default='water', choices=['water', 'taxol'])
ap.add_argument('--world', type=int, default=1)
ap.add_argument('--rank', type=int, default=0)
a = ap.parse_args()
mol = gto.M(atom=clusters.taxol() if a.molecule == 'taxol' else clusters.water_cluster(a.nwater), basis=a.basis)
nao = mol.nao
obj = df.DF(mol)
obj.k_square = False
if a.world > 1:
    obj._shard_override = (a.rank, a.world)
obj.build()
cd = obj._cderi_dev
nL, npair = cd.shape
mx = torch.zeros(npair, dtype=torch.float64, device=cd.device)
for b0 in range(0, nL, 64):
    mx = torch.maximum(mx, cd[b0:b0 + 64].abs().amax(0))
mxh = mx.cpu().numpy()
M = np.zeros((nao, nao))
M[np.tril_indices(nao)] = mxh
M = np.maximum(M, M.T)
scale = float(mxh.max())


def tiles(M, tq, tp):
    nq, np_ = -(-nao // tq) * tq, -(-nao // tp) * tp
    P = np.zeros((nq, np_))
    P[:nao, :nao] = M
    return P.reshape(nq // tq, tq, np_ // tp, tp).max(axis=(1, 3))


out = {'molecule': a.molecule if a.molecule != 'water' else '(H2O)_%d' % a.nwater, 'basis': a.basis, 'nao': nao, 'rows': int(nL),
       'shard': [a.rank, a.world], 'max_abs': scale, 'live_fraction': {}}
t128, t16 = tiles(M, 16, 128), tiles(M, 16, 16)
for thr in (1e-13, 1e-12, 1e-10, 1e-8):
    out['live_fraction']['%g' % thr] = {
        'k_tiles_16q_x_128p': round(float((t128 > thr).mean()), 4),
        'tiles_16x16': round(float((t16 > thr).mean()), 4),
        'pairs': round(float((M > thr).mean()), 4),
        # per column tile: the worst (densest) and best (sparsest) column tile's live share of its k-tiles
        'per_column_tile_min_max': [round(float((t128 > thr).mean(axis=0).min()), 4), round(float((t128 > thr).mean(axis=0).max()), 4)],
    }
print(json.dumps(out))
