"""Oracle-only DF-RHF ENERGY golden at a size whose tensor (and even whose half-transformed tensor) fits no CPU box:
BASELINE config 5, (H2O)_128 cc-pVDZ - nao 3072, naux 14 848, nocc 640: B[L,pq] is 560 GB, X[L,i,p] 233 GB.

What makes it affordable is that the ENERGY functional (pyscf/scf/hf.py:268-300, with J/K of pyscf/df/df_jk.py:280-381) needs the
tensor only through occupied-occupied quantities.  With D = C C^T (C = C_occ sqrt(2)), M = (P|Q) = L L^T and the RAW integrals T:

    gamma[Q]   = sum_pq T[Q,pq] D[pq]                          rho = L^-1 gamma         1/2 Tr(D J) = 1/2 rho.rho
    Z[Q,i,j]   = sum_pq C[p,i] T[Q,pq] C[q,j]  (24 GB, i >= j)  Y   = L^-1 Z             Tr(D K)     = sum_Lij Y[L,i,j]^2
    E[D] = Tr(h D) + 1/2 rho.rho - 1/4 sum Y^2 + E_nuc

so ONE sweep of the oracle's McMurchie-Davidson integrals (oracle/cint_oracle.c), AO-row slab by slab, no triangular solve of any
slab, gives the oracle's energy at the density of --orbitals (a converged product SCF).  Stationarity is then checked BY THE ORACLE
on sampled AO rows P (all functions of two water molecules, the most central and the outermost): with the rectangular integrals
(Q|pq), p in P, all q,

    J[p,q]    = sum_Q u[Q] (Q|pq),  u = L^-T rho                (K C)[p,i] = sum_{L,k} (L|pk) Y[L,k,i],  (L|pk) = L^-1 (Q|pq) C[q,k]
    R[p,:]    = (F C)[p,:] - (S C)[p,:] (C^T F C) / 2           rows of the Roothaan residual; C^T F C from h, u.Z and Y Y

|R| ~ the orbital gradient: the energy lies above the oracle's own minimum by O(|R|^2).  The J rows and (K C) rows are stored as
sampled goldens too (a product J/K build at the same orbitals is compared element-wise).

    python tools/gen_golden_energy_sweep.py --orbitals gpurun_out/cfg5scf/h2o128_rhf_orbitals.npz          (~2-3 h on 8 cores)
    python tools/gen_golden_energy_sweep.py --nwater 4 --selfcheck                                          (in-core comparison)
Writes tests/golden/h2o<N>_<basis>_energy_oracle.json and copies the orbitals next to it.
"""
import argparse
import json
import os
import shutil
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
ap.add_argument('--orbitals', default='', help='.npz with orbo (nao, nocc) = C_occ sqrt(2) of a converged SCF and e_tot')
ap.add_argument('--selfcheck', action='store_true', help='small case: orbitals from the oracle\'s own in-core SCF, every quantity '
                'compared with the in-core oracle (ref.cholesky_eri / ref.get_jk)')
ap.add_argument('--slab-bytes', type=float, default=3.0e9)
ap.add_argument('--super-rows', type=int, default=96, help='AO rows whose half transform is held before the second index is contracted')
ap.add_argument('--nblk', type=int, default=8, help='orbital blocks of the (i >= j) block-triangular Z storage')
ap.add_argument('--nsample', type=int, default=2048)
ap.add_argument('--tag', default='')
a = ap.parse_args()
t00 = time.time()


def log(*args):
    print('[%7.1fs]' % (time.time() - t00), *args, flush=True)


mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
auxmol = addons.make_auxmol(mol, None)
nao, naux = mol.nao, auxmol.nao_nr()
npair = nao * (nao + 1) // 2
tag = a.tag or 'h2o%d_%s_energy' % (a.nwater, a.basis.replace('-', ''))
out_json = os.path.join(ROOT, 'tests', 'golden', tag + '_oracle.json')
loc = ref.ao_loc(mol)
h1e = ref.int1e(mol, 'kin') + ref.int1e(mol, 'nuc')
s1e = ref.int1e(mol, 'ovlp')

cderi_chk = None
if a.selfcheck:
    cderi_chk = ref.cholesky_eri(mol, auxmol)

    def veff(dm, *args, **kw):
        vj, vk = ref.get_jk(cderi_chk, dm, 1)
        return vj - .5 * vk
    conv, e_src, mo_e, mo_c, mo_occ = ref.rhf_kernel(mol, veff, conv_tol=1e-11)[:5]
    orbo = np.ascontiguousarray(mo_c[:, mo_occ > 0] * np.sqrt(2.0))
    src = 'the oracle\'s own in-core SCF (selfcheck)'
else:
    z = np.load(a.orbitals)
    orbo = np.ascontiguousarray(z['orbo'])
    e_src = float(z['e_tot'])
    src = os.path.basename(a.orbitals)
nocc = orbo.shape[1]
assert orbo.shape[0] == nao
log('nao', nao, 'naux', naux, 'nocc', nocc, 'tensor GB %.1f' % (8e-9 * naux * npair), 'orbitals from', src, 'E there %.12f' % e_src)

dm = orbo.dot(orbo.T)
dtril = ref.pack_tril(dm + dm.T)
idx = np.arange(nao)
dtril[idx * (idx + 1) // 2 + idx] *= .5

j2c = ref.int2c2e(auxmol)
low = scipy.linalg.cholesky(j2c, lower=True)
log('metric factorised, fp(j2c) %.12f' % golden_util.fp(j2c))

# block-triangular storage of Z[Q, i, j], i-block >= j-block
nb = a.nblk
edges = [nocc * k // nb for k in range(nb + 1)]
Zb = {(x, y): np.zeros((naux, edges[x + 1] - edges[x], edges[y + 1] - edges[y])) for x in range(nb) for y in range(x + 1)}
log('Z blocks GB %.1f' % (sum(v.nbytes for v in Zb.values()) * 1e-9))
gamma = np.zeros(naux)


def slabs():
    ish0 = 0
    while ish0 < mol.nbas:
        ish1 = ish0 + 1

        def ncol(s0, s1):
            return loc[s1] * (loc[s1] + 1) // 2 - loc[s0] * (loc[s0] + 1) // 2
        while ish1 < mol.nbas and ncol(ish0, ish1 + 1) * naux * 8 <= a.slab_bytes:
            ish1 += 1
        yield ish0, ish1
        ish0 = ish1


QC = 256


def flush(vh, rows):
    """second index: M[Q,i,j] = sum_{p in rows} C[p,i] vh[Q,p,j];  Z += M + M^T (block-triangular)."""
    cp = np.ascontiguousarray(orbo[rows[0]:rows[1]].T)            # (nocc, R)
    for q0 in range(0, naux, QC):
        q1 = min(q0 + QC, naux)
        m = np.matmul(cp[None], vh[q0:q1, :rows[1] - rows[0]])   # (qc, nocc, nocc)
        for (x, y), zb in Zb.items():
            i0, i1, j0, j1 = edges[x], edges[x + 1], edges[y], edges[y + 1]
            zb[q0:q1] += m[:, i0:i1, j0:j1]
            zb[q0:q1] += m[:, j0:j1, i0:i1].transpose(0, 2, 1)


# ------------------------------------------------------------------------------------------- the sweep
R = a.super_rows
vh = np.zeros((naux, R, nocc))
r0 = 0                    # first AO row held in vh
t_int = t_g1 = t_g2 = 0.0
for ish0, ish1 in slabs():
    p0, p1 = int(loc[ish0]), int(loc[ish1])
    t = time.time()
    T = ref.int3c2e_slab(mol, auxmol, ish0, ish1)                 # (naux, ncol): rows p of the slab, q <= p
    t_int += time.time() - t
    pq0 = p0 * (p0 + 1) // 2
    gamma += T.dot(dtril[pq0:pq0 + T.shape[1]])
    for p in range(p0, p1):
        if p - r0 == R:
            t = time.time()
            flush(vh, (r0, p))
            t_g2 += time.time() - t
            r0 = p
        off = p * (p + 1) // 2 - pq0
        t = time.time()
        v = T[:, off:off + p + 1].dot(orbo[:p + 1])
        v -= .5 * np.outer(T[:, off + p], orbo[p])               # the diagonal pair counts once in M + M^T
        vh[:, p - r0] = v
        t_g1 += time.time() - t
    del T
    log('slab shells [%d,%d) rows [%d,%d): integrals %.0f s, first index %.0f s, second index %.0f s (cumulative)' % (
        ish0, ish1, p0, p1, t_int, t_g1, t_g2))
flush(vh, (r0, nao))
del vh
log('sweep done')

# ------------------------------------------------------------------------------------------- energies
rho = scipy.linalg.solve_triangular(low, gamma, lower=True)
u = scipy.linalg.solve_triangular(low, rho, lower=True, trans='T')
e_coul = .5 * float(rho.dot(rho))
# sum_Q u_Q Z[Q] (J in the occupied basis) before Z is overwritten
j_oo = np.zeros((nocc, nocc))
for (x, y), zb in Zb.items():
    blk = np.tensordot(u, zb, axes=(0, 0))
    j_oo[edges[x]:edges[x + 1], edges[y]:edges[y + 1]] = blk
    if x != y:
        j_oo[edges[y]:edges[y + 1], edges[x]:edges[x + 1]] = blk.T
sumy2 = 0.0
for (x, y), zb in Zb.items():
    t = time.time()
    flat = zb.reshape(naux, -1)
    for c0 in range(0, flat.shape[1], 4096):                      # column panels: the solve runs in place, panel by panel
        flat[:, c0:c0 + 4096] = scipy.linalg.solve_triangular(low, flat[:, c0:c0 + 4096], lower=True, check_finite=False)
    sumy2 += (1.0 if x == y else 2.0) * float(np.vdot(flat, flat))
    log('Y block', (x, y), '%.0f s' % (time.time() - t))
e_x = -.25 * sumy2
e_1 = float(np.einsum('ij,ji', h1e, dm))
e_nuc = float(mol.energy_nuc())
e_tot = e_1 + e_coul + e_x + e_nuc
log('E_oracle[D] = %.12f   (E1 %.12f  Ecoul %.12f  Ex %.12f  Enuc %.12f);  source reported %.12f,  diff %.3e' % (
    e_tot, e_1, e_coul, e_x, e_nuc, e_src, e_tot - e_src))

res = {'system': '(H2O)_%d %s (aux by the reference rule, %d functions), DF-RHF' % (a.nwater, a.basis, naux),
       'generator': 'tools/gen_golden_energy_sweep.py (CPU oracle only: oracle/cint_oracle.c integrals, scipy Cholesky / trsm)',
       'nao': nao, 'naux': naux, 'nocc': nocc, 'orbital_source': src, 'e_tot_of_the_orbital_source': e_src,
       'e_tot': e_tot, 'e1': e_1, 'e_coul': e_coul, 'e_x': e_x, 'e_nuc': e_nuc, 'j2c_fp': golden_util.fp(j2c),
       'nelec': float(np.einsum('ij,ji', dm, s1e)),
       'orthonormality': float(np.abs(orbo.T.dot(s1e).dot(orbo) - 2 * np.eye(nocc)).max())}
with open(out_json, 'w') as f:
    json.dump(res, f, indent=1)
log('energy written', out_json)

# ------------------------------------------------------------------------------------------- stationarity on sampled rows
# occupied-occupied Fock (scaled orbitals): C^T F C = C^T h C + j_oo - 1/2 sum_L Y_L Y_L
k_oo = np.zeros((nocc, nocc))
for q0 in range(0, naux, QC):
    q1 = min(q0 + QC, naux)
    yfull = np.zeros((q1 - q0, nocc, nocc))
    for (x, y), zb in Zb.items():
        yfull[:, edges[x]:edges[x + 1], edges[y]:edges[y + 1]] = zb[q0:q1]
        if x != y:
            yfull[:, edges[y]:edges[y + 1], edges[x]:edges[x + 1]] = zb[q0:q1].transpose(0, 2, 1)
    k_oo += yfull.transpose(1, 0, 2).reshape(nocc, -1).dot(yfull.reshape(-1, nocc))
f_oo = orbo.T.dot(h1e).dot(orbo) + j_oo - .5 * k_oo
res['e_tot_from_occupied_fock'] = float(.5 * (np.trace(orbo.T.dot(h1e).dot(orbo)) + np.trace(f_oo))) + e_nuc

coords = mol.atom_coords()
cen = coords[::3].mean(axis=0)                                   # oxygens (O, H, H order per molecule)
dist = np.linalg.norm(coords[::3] - cen, axis=1)
picks = [int(np.argmin(dist)), int(np.argmax(dist))]
atom_of = np.asarray(mol._bas)[:, 0]
sc = s1e.dot(orbo)
rr, cc = golden_util.sample_positions(1 << 20, a.nsample)
res['row_samples'] = []
rnorm2 = 0.0
for m in picks:
    shells = np.flatnonzero((atom_of >= 3 * m) & (atom_of < 3 * m + 3))
    assert len(shells) and shells[-1] - shells[0] + 1 == len(shells)
    s0, s1 = int(shells[0]), int(shells[-1]) + 1
    p0, p1 = int(loc[s0]), int(loc[s1])
    t = time.time()
    tp = ref.int3c2e_block(mol, auxmol, s0, s1, 0, mol.nbas)     # (naux, np, nao)
    npp = p1 - p0
    jrows = np.tensordot(u, tp, axes=(0, 0))                     # (np, nao)
    half = tp.reshape(-1, nao).dot(orbo).reshape(naux, npp * nocc)        # (Q | p k)
    del tp
    half = scipy.linalg.solve_triangular(low, half, lower=True, overwrite_b=True, check_finite=False).reshape(naux, npp, nocc)
    kc = np.zeros((npp, nocc))
    for q0 in range(0, naux, QC):
        q1 = min(q0 + QC, naux)
        yfull = np.zeros((q1 - q0, nocc, nocc))
        for (x, y), zb in Zb.items():
            yfull[:, edges[x]:edges[x + 1], edges[y]:edges[y + 1]] = zb[q0:q1]
            if x != y:
                yfull[:, edges[y]:edges[y + 1], edges[x]:edges[x + 1]] = zb[q0:q1].transpose(0, 2, 1)
        kc += half[q0:q1].transpose(1, 0, 2).reshape(npp, -1).dot(yfull.reshape(-1, nocc))
    del half
    fc = h1e[p0:p1].dot(orbo) + jrows.dot(orbo) - .5 * kc
    resid = fc - sc[p0:p1].dot(f_oo) * .5
    rnorm2 += float(np.vdot(resid, resid))
    ri, ci = rr % npp, cc % nao
    rk, ck = rr % npp, cc % nocc
    res['row_samples'].append({
        'molecule': m, 'ao_rows': [p0, p1], 'distance_from_centroid_bohr': float(dist[m]),
        'vj_rows_fp': golden_util.fp(jrows), 'vj_rows_norm': float(np.linalg.norm(jrows)), 'vj_rows_absmax': float(np.abs(jrows).max()),
        'vj_rows_sample': [float(v) for v in jrows[ri, ci]],
        'vkc_rows_fp': golden_util.fp(kc), 'vkc_rows_norm': float(np.linalg.norm(kc)), 'vkc_rows_absmax': float(np.abs(kc).max()),
        'vkc_rows_sample': [float(v) for v in kc[rk, ck]],
        'roothaan_residual_norm': float(np.linalg.norm(resid)), 'roothaan_residual_absmax': float(np.abs(resid).max())})
    log('molecule %d rows [%d,%d): |R| %.3e  max %.3e  (%.0f s)' % (m, p0, p1, np.linalg.norm(resid), np.abs(resid).max(), time.time() - t))
    if cderi_chk is not None:
        vj0, vk0 = ref.get_jk(cderi_chk, dm, 1)
        log('   selfcheck rows: |J - J0| %.2e  |KC - K0 C| %.2e' % (np.abs(jrows - vj0[p0:p1]).max(), np.abs(kc - vk0[p0:p1].dot(orbo)).max()))
res['sample_seed'] = 11
res['roothaan_residual_norm_sampled_rows'] = float(np.sqrt(rnorm2))
res['note'] = ("e_tot: the oracle's DF-RHF energy functional at the density of the orbital source; roothaan_residual_*: "
               "|F C - S C (C^T F C)/2| on the AO rows of two molecules, by the oracle; the energy is stationary to O(|R|^2)")
with open(out_json, 'w') as f:
    json.dump(res, f, indent=1)
if cderi_chk is not None:
    vj0, vk0 = ref.get_jk(cderi_chk, dm, 1)
    e0 = float(np.einsum('ij,ji', h1e + .5 * (vj0 - .5 * vk0), dm)) + e_nuc
    log('selfcheck: E in core %.12f, sweep %.12f, diff %.2e; occupied-Fock route diff %.2e' % (e0, e_tot, e_tot - e0, res['e_tot_from_occupied_fock'] - e0))
    assert abs(e_tot - e0) < 1e-9 and abs(res['e_tot_from_occupied_fock'] - e0) < 1e-9
elif a.orbitals:
    dst = os.path.join(ROOT, 'tests', 'golden', tag.replace('_energy', '') + '_rhf_orbitals.npz')
    if os.path.abspath(a.orbitals) != dst:
        shutil.copy(a.orbitals, dst)
log('written', out_json)
