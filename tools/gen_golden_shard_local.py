"""Golden partial J/K of ONE RANK'S aux-row shard at BASELINE config 5 ((H2O)_128 cc-pVDZ: nao 3072, naux 14 848, a 560 GB
tensor nobody can generate on the CPU box), computed by the CPU oracle alone.

Trick: the density is LOCAL - D = C~ C~^T with orbitals supported on the AOs S of `--local-waters` molecules, by default the ones
whose aux functions open the rank's row range (so that the partial J/K are of order one, not the far tail of another region).
Then, for the rank's aux rows L in [l0, l1):

    rho_L        = sum_{p,q in S} B[L,pq] D_pq
    X[L,i,p]     = sum_{q in S}   B_L[p,q] C~[q,i]                     (every p)
    K_part[p,p'] = sum_{L,i} X[L,i,p] X[L,i,p']                         (the FULL nao x nao matrix)
    J_part[p,q]  = sum_L rho_L B_L[p,q]                                  (p any, q in S: the rectangle the block covers)

need only the integral block (Q | p q), p any, q in S, for the aux functions Q < l1 (rows [l0, l1) of the lower-triangular L^-1
vanish beyond): l1 x nao x |S| integrals (4.4e9 for rank 3 of 8 and 8 molecules) instead of naux x nao_pair (7e10).  The product
is asked for the same thing through its ordinary kernels: the rank's shard of the REAL tensor (70 GB on the GPU), get_jk_device
with this density (`DF._shard_override = (rank, world)`: no collective).  What is checked is every tensor row of the shard
against the oracle's integrals and Cholesky factor, contracted by the production kernels.

    python tools/gen_golden_shard_local.py --nwater 128 --basis cc-pvdz --rank 3 --world 8 --local-waters 8 --nsyn 32

Writes tests/golden/h2o<n>_<basis>_rank<r>of<w>_local_oracle.json (+ .npz with the J rectangle and K samples).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref, golden_util          # noqa: E402
from pyscf_amd import gto                     # noqa: E402  (host-only: molecule tables, basis data)
from pyscf_amd.data import clusters           # noqa: E402
from pyscf_amd.df import addons               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=128)
ap.add_argument('--basis', default='cc-pvdz')
ap.add_argument('--rank', type=int, default=3)
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--local-waters', type=int, default=8)
ap.add_argument('--first-water', type=int, default=-1, help='first molecule of the support (default: the molecule whose aux '
                'functions open the rank\'s row range, so that the partial J/K are of order one, not a far tail)')
ap.add_argument('--nsyn', type=int, default=32)
ap.add_argument('--rows', type=int, nargs=2, default=None, help='explicit aux row range [l0, l1) instead of a rank (e.g. two adjacent shards at once)')
ap.add_argument('--tag', default='')
ap.add_argument('--rows-per-pass', type=int, default=0, help='AO shells per integral block (0: sized for ~4 GB)')
ap.add_argument('--nsample', type=int, default=4096)
ap.add_argument('--check-dense', action='store_true', help='small cases: compare with the dense oracle tensor')
a = ap.parse_args()
t00 = time.time()


def log(*args):
    print('[%7.1fs]' % (time.time() - t00), *args, flush=True)


mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
auxmol = addons.make_auxmol(mol, None)
nao, naux = mol.nao, auxmol.nao_nr()
base, rem = divmod(naux, a.world)                      # DF.shard_range
l0 = a.rank * base + min(a.rank, rem)
l1 = l0 + base + (1 if a.rank < rem else 0)
if a.rows:
    l0, l1 = a.rows
nl = l1 - l0
loc = ref.ao_loc(mol)
if a.first_water < 0:
    aux_per_water = naux // a.nwater
    a.first_water = min(l0 // aux_per_water, a.nwater - a.local_waters)
atm0, atm1 = 3 * a.first_water, 3 * (a.first_water + a.local_waters)
sup = [i for i in range(mol.nbas) if atm0 <= mol._bas[i, 0] < atm1]
jsh0, jsh1 = sup[0], sup[-1] + 1
assert sup == list(range(jsh0, jsh1)), 'the shells of the support molecules must be contiguous'
a0, a1 = int(loc[jsh0]), int(loc[jsh1])
ns = a1 - a0                                           # |S|: the support is the AO range [a0, a1)
c = np.zeros((nao, a.nsyn))
c[a0:a1] = golden_util.synthetic_orbitals(ns, a.nsyn) * np.sqrt(2.0)
dm_ss = c[a0:a1].dot(c[a0:a1].T)

j2c = ref.int2c2e(auxmol)
low = scipy.linalg.cholesky(j2c, lower=True)
linv_rows = np.ascontiguousarray(scipy.linalg.solve_triangular(low, np.eye(naux), lower=True)[l0:l1])
# rows [l0, l1) of the lower-triangular L^-1 vanish for Q >= l1: only the aux shells below l1 are needed
aloc = ref.ao_loc(auxmol)
nbas_aux_used = int(np.searchsorted(aloc, l1, 'left'))
naux_used = int(aloc[nbas_aux_used])
assert naux_used >= l1 and (naux_used == naux or np.abs(linv_rows[:, naux_used:]).max() == 0)
linv_rows = np.ascontiguousarray(linv_rows[:, :naux_used])


class _AuxView:                                        # the first nbas_aux_used shells of the aux basis
    pass


auxv = _AuxView()
auxv._atm, auxv._env = auxmol._atm, auxmol._env
auxv._bas = np.asarray(auxmol._bas)[:nbas_aux_used]
auxv.nbas = nbas_aux_used
auxv.nao_nr = lambda: naux_used
log('nao', nao, 'naux', naux, 'aux rows [%d, %d)' % (l0, l1), 'support AOs [%d, %d)' % (a0, a1), 'aux functions used', naux_used,
    'integrals %.2e' % (float(naux_used) * nao * ns))
log('metric factorised, fp(j2c) %.12f' % golden_util.fp(j2c))

# B_S[L, p, q in S] for the rank's rows: (nl, nao, ns)
BS = np.empty((nl, nao, ns))
log('B_S GB', BS.nbytes * 1e-9)
nsh_blk = a.rows_per_pass or max(1, int(4e9 // (8.0 * naux_used * ns * 6)))       # ~6 functions per shell
ish0 = 0
while ish0 < mol.nbas:
    ish1 = min(mol.nbas, ish0 + nsh_blk)
    t = time.time()
    blk = ref.int3c2e_block(mol, auxv, ish0, ish1, jsh0, jsh1)                # (naux_used, np, ns)
    p0, p1 = int(loc[ish0]), int(loc[ish1])
    BS[:, p0:p1] = linv_rows.dot(blk.reshape(naux_used, -1)).reshape(nl, p1 - p0, ns)
    log('shells [%d,%d) rows [%d,%d) %.1f s' % (ish0, ish1, p0, p1, time.time() - t))
    ish0 = ish1

rho = np.einsum('Lpq,pq->L', BS[:, a0:a1], dm_ss)
X = BS.reshape(-1, ns).dot(c[a0:a1]).reshape(nl, nao, a.nsyn)                   # X[L, p, i]
x2 = X.transpose(0, 2, 1).reshape(-1, nao)
vk = x2.T.dot(x2)
vj_rect = np.einsum('L,Lpq->pq', rho, BS)                                     # J_part[p, q in S]
log('contracted')

if a.check_dense:
    cd = ref.cholesky_eri(mol, auxmol)[l0:l1]
    dm = c.dot(c.T)
    vj0, vk0 = ref.get_jk(cd, dm, 1)
    print('check vs dense oracle: |dJ| %.2e |dK| %.2e' % (np.abs(vj0[:, a0:a1] - vj_rect).max(), np.abs(vk0 - vk).max()))

tag = a.tag or ('h2o%d_%s_rows%d-%d_local' % (a.nwater, a.basis.replace('-', ''), l0, l1) if a.rows else
                'h2o%d_%s_rank%dof%d_local' % (a.nwater, a.basis.replace('-', ''), a.rank, a.world))
ri, ci = golden_util.sample_positions(nao, a.nsample)
res = {'system': '(H2O)_%d %s, aux rows [%d, %d) of %d (%s)' % (a.nwater, a.basis, l0, l1, naux, 'explicit row range' if a.rows else 'rank %d of %d' % (a.rank, a.world)),
       'nao': nao, 'naux': naux, 'aux_rows': [l0, l1], 'support_ao_range': [a0, a1], 'support_waters': [a.first_water,
                                                                                                 a.first_water + a.local_waters],
       'nsyn': a.nsyn,
       'density': 'D = C C^T, C[%d:%d] = sqrt(2) oracle.golden_util.synthetic_orbitals(%d, %d), other rows 0' % (a0, a1, ns, a.nsyn),
       'generator': 'tools/gen_golden_shard_local.py (CPU oracle only)',
       'j2c_fp': golden_util.fp(j2c),
       'vk_fp': golden_util.fp(vk), 'vk_norm': float(np.linalg.norm(vk)), 'vk_absmax': float(np.abs(vk).max()),
       'vj_rect_fp': golden_util.fp(vj_rect), 'vj_rect_norm': float(np.linalg.norm(vj_rect)),
       'vj_rect_absmax': float(np.abs(vj_rect).max()),
       'rho_fp': golden_util.fp(rho), 'rho_norm': float(np.linalg.norm(rho)),
       'sample_seed': 11, 'vk_sample': [float(v) for v in vk[ri, ci]],
       'vj_rect_sample': [float(v) for v in vj_rect[ri, ci % ns]]}
with open(os.path.join(ROOT, 'tests', 'golden', tag + '_oracle.json'), 'w') as f:
    json.dump(res, f, indent=1)
log('written', tag, 'fp(vk) %.12f fp(vj_rect) %.12f' % (res['vk_fp'], res['vj_rect_fp']))
