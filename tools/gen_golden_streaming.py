"""Golden J/K vectors and energies at sizes whose tensor does not fit the CPU box (taxol def2-TZVP: 111 GB; one rank's view of
(H2O)_128 cc-pVDZ: 560 GB), computed by the CPU oracle ALONE without ever holding the tensor: the oracle's own
McMurchie-Davidson integrals (oracle/cint_oracle.c) are generated AO-row slab by slab, decomposed with the oracle's own
Cholesky factor (pyscf/df/incore.py:129-220) and contracted on the fly, in two passes over the slabs:

  pass 1   rho[s][L] += B[L, slab] . dtril[s][slab]                                   (J, first half; df_jk.py:367)
           X[s][L][i][p] += sum_q B_L[p,q] C_s[q,i]   for the (p,q) pairs of the slab    (half transform, nr_ao2mo.c:399-419)
  K[s] = sum_{L,i} X[s][L][i][:]^T X[s][L][i][:]                                       (df_jk.py:380)
  pass 2   J~[s][slab] = rho[s] . B[:, slab]                                            (J, second half)

Only rows [l0, l1) of the aux index are contracted when --rank/--world select a shard (B = (L^-1)[l0:l1, :] (Q|pq)): that is
one rank's partial J/K of the aux-sharded build.

    python tools/gen_golden_streaming.py --molecule taxol --orbitals gpurun_out/taxol_rhf_orbitals.npz
    python tools/gen_golden_streaming.py --molecule water --nwater 16 --basis cc-pvtz --rank 1 --world 2     (small-case self-check)

Densities: 'syn' = D = 2 C C^T with C = oracle.golden_util.synthetic_orbitals(nao, nsyn) (a seeded density any test can
rebuild); 'conv' = the occupied orbitals handed over by --orbitals (a converged product SCF): the oracle then reports ITS energy
functional E[D] = Tr(hD) + 1/2 Tr(D (J - K/2)) + E_nuc and ITS orbital gradient norm at that density - an energy above the
oracle's own minimum by O(|g|^2).  (One rank's shard of a 560 GB-class tensor: tools/gen_golden_shard_local.py.)
Writes tests/golden/<tag>_oracle.json.
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
ap.add_argument('--molecule', default='taxol', choices=['taxol', 'water'])
ap.add_argument('--nwater', type=int, default=128)
ap.add_argument('--basis', default=None)
ap.add_argument('--orbitals', default='', help='.npz with orbo (nao, nocc) = C_occ sqrt(2) of a converged SCF and e_tot')
ap.add_argument('--nsyn', type=int, default=32, help='rank of the synthetic density (0: none)')
ap.add_argument('--rank', type=int, default=0)
ap.add_argument('--world', type=int, default=1)
ap.add_argument('--slab-bytes', type=float, default=2.5e9)
ap.add_argument('--nsample', type=int, default=4096)
ap.add_argument('--tag', default='')
a = ap.parse_args()
t00 = time.time()


def log(*args):
    print('[%7.1fs]' % (time.time() - t00), *args, flush=True)


if a.basis is None:
    a.basis = 'def2-tzvp' if a.molecule == 'taxol' else 'cc-pvdz'
atoms = clusters.taxol() if a.molecule == 'taxol' else clusters.water_cluster(a.nwater)
mol = gto.M(atom=atoms, basis=a.basis)
auxmol = addons.make_auxmol(mol, None)
nao, naux = mol.nao, auxmol.nao_nr()
npair = nao * (nao + 1) // 2
tag = a.tag or ('taxol_%s' % a.basis.replace('-', '') if a.molecule == 'taxol'
                else 'h2o%d_%s' % (a.nwater, a.basis.replace('-', '')))
if a.world > 1:
    tag += '_rank%dof%d' % (a.rank, a.world)
out_json = os.path.join(ROOT, 'tests', 'golden', tag + '_oracle.json')
res = json.load(open(out_json)) if os.path.exists(out_json) else {}
base, rem = divmod(naux, a.world)                      # DF.shard_range
l0 = a.rank * base + min(a.rank, rem)
l1 = l0 + base + (1 if a.rank < rem else 0)
nl = l1 - l0
log('nao', nao, 'naux', naux, 'npair', npair, 'tensor GB', 8e-9 * naux * npair, 'aux rows [%d, %d)' % (l0, l1))
res.update({'system': '%s %s (aux by the reference rule, %d functions)' % (
                'taxol C47H51NO14 (pyscf_amd/data/taxol.xyz)' if a.molecule == 'taxol' else '(H2O)_%d' % a.nwater, a.basis, naux),
            'nao': nao, 'naux': naux, 'aux_rows': [l0, l1],
            'generator': 'tools/gen_golden_streaming.py (CPU oracle only: oracle/cint_oracle.c integrals, scipy Cholesky / trsm)'})

# ------------------------------------------------------------------------------------------- densities
loc = ref.ao_loc(mol)
sets = []            # (name, C (nao, r) with D = C C^T, support rows)
if a.nsyn:
    c = golden_util.synthetic_orbitals(nao, a.nsyn)
    sets.append(('syn', c * np.sqrt(2.0), 'D = 2 C C^T, C = oracle.golden_util.synthetic_orbitals(nao, %d)' % a.nsyn))
if a.orbitals:
    z = np.load(a.orbitals)
    sets.append(('conv', np.ascontiguousarray(z['orbo']), 'occupied orbitals of %s (E = %.12f there)' % (
        os.path.basename(a.orbitals), float(z['e_tot']))))
    res['conv_e_tot_of_the_orbital_source'] = float(z['e_tot'])
assert sets
log('densities:', [(n, c.shape[1]) for n, c, _ in sets])
dms = [c.dot(c.T) for _, c, _ in sets]
dtrils = []
for d in dms:
    t = ref.pack_tril(d + d.T)
    idx = np.arange(nao)
    t[idx * (idx + 1) // 2 + idx] *= .5
    dtrils.append(t)

# ------------------------------------------------------------------------------------------- metric
j2c = ref.int2c2e(auxmol)
low = scipy.linalg.cholesky(j2c, lower=True)
res['j2c_fp'] = golden_util.fp(j2c)
# rows [l0, l1) of L^-1: B[l0:l1] = linv_rows @ T
linv_rows = np.ascontiguousarray(scipy.linalg.solve_triangular(low, np.eye(naux), lower=True)[l0:l1]) if a.world > 1 else None
log('metric factorised')


def slabs():
    """AO-row shell slabs [ish0, ish1) bounded by --slab-bytes."""
    ish0 = 0
    while ish0 < mol.nbas:
        ish1 = ish0 + 1

        def ncol(s0, s1):
            return loc[s1] * (loc[s1] + 1) // 2 - loc[s0] * (loc[s0] + 1) // 2
        while ish1 < mol.nbas and ncol(ish0, ish1 + 1) * naux * 8 <= a.slab_bytes:
            ish1 += 1
        yield ish0, ish1
        ish0 = ish1


def slab_tensor(ish0, ish1):
    t = ref.int3c2e_slab(mol, auxmol, ish0, ish1)                       # (naux, ncol) raw (Q|pq), packed rows of the slab
    if linv_rows is None:
        return scipy.linalg.solve_triangular(low, t, lower=True, overwrite_b=True, check_finite=False)
    return linv_rows.dot(t)


# ------------------------------------------------------------------------------------------- pass 1
rhos = [np.zeros(nl) for _ in sets]
X = [np.zeros((nl, c.shape[1], nao)) for _, c, _ in sets]
log('X buffers GB', sum(x.nbytes for x in X) * 1e-9)
for ish0, ish1 in slabs():
    p0, p1 = loc[ish0], loc[ish1]
    t = time.time()
    b = slab_tensor(ish0, ish1)
    pq0 = p0 * (p0 + 1) // 2
    for s, (name, c, _) in enumerate(sets):
        rhos[s] += b.dot(dtrils[s][pq0:pq0 + b.shape[1]])
    # symmetric slab image S[L, p - p0, q] for q < p1 (q > p inside the slab filled by symmetry)
    S = np.zeros((nl, p1 - p0, p1))
    for p in range(p0, p1):
        off = p * (p + 1) // 2 - pq0
        S[:, p - p0, :p + 1] = b[:, off:off + p + 1]
    for p in range(p0, p1):
        S[:, p - p0, p + 1:p1] = S[:, p + 1 - p0:p1 - p0, p]
    for s, (name, c, _) in enumerate(sets):
        # X[L, i, p in slab] += sum_{q < p1} S[L, p, q] C[q, i]
        r = c.shape[1]
        X[s][:, :, p0:p1] += S.reshape(-1, p1).dot(c[:p1]).reshape(nl, p1 - p0, r).transpose(0, 2, 1)
        # X[L, i, q < p0] += sum_{p in slab} S[L, p, q] C[p, i]
        if p0:
            X[s][:, :, :p0] += np.matmul(c[p0:p1].T[None], S[:, :, :p0])
    del S, b
    log('pass 1 slab shells [%d,%d) rows [%d,%d) %.1f s' % (ish0, ish1, p0, p1, time.time() - t))

ri, ci = golden_util.sample_positions(nao, a.nsample)
vks = []
for s, (name, c, desc) in enumerate(sets):
    t = time.time()
    x2 = X[s].reshape(-1, nao)
    vk = x2.T.dot(x2)
    vks.append(vk)
    log('K[%s] %.1f s' % (name, time.time() - t))
del X

# ------------------------------------------------------------------------------------------- pass 2 (J)
vjt = [np.zeros(npair) for _ in sets]
for ish0, ish1 in slabs():
    p0, p1 = loc[ish0], loc[ish1]
    t = time.time()
    b = slab_tensor(ish0, ish1)
    pq0 = p0 * (p0 + 1) // 2
    for s in range(len(sets)):
        vjt[s][pq0:pq0 + b.shape[1]] = rhos[s].dot(b)
    log('pass 2 slab shells [%d,%d) %.1f s' % (ish0, ish1, time.time() - t))

h1e = s1e = None
for s, (name, c, desc) in enumerate(sets):
    vj = ref.unpack_tril(vjt[s])
    vk = vks[s]
    dm = dms[s]
    key = name + '_'
    res.update({key + 'density': desc,
                key + 'vj_fp': golden_util.fp(vj), key + 'vk_fp': golden_util.fp(vk),
                key + 'vj_norm': float(np.linalg.norm(vj)), key + 'vk_norm': float(np.linalg.norm(vk)),
                key + 'vj_absmax': float(abs(vj).max()), key + 'vk_absmax': float(abs(vk).max()),
                key + 'tr_d_vj': float(np.einsum('ij,ji', dm, vj)), key + 'tr_d_vk': float(np.einsum('ij,ji', dm, vk)),
                'sample_seed': 11, key + 'vj_sample': [float(v) for v in vj[ri, ci]],
                key + 'vk_sample': [float(v) for v in vk[ri, ci]]})
    if name == 'conv' and a.world == 1:
        h1e = ref.int1e(mol, 'kin') + ref.int1e(mol, 'nuc')
        s1e = ref.int1e(mol, 'ovlp')
        vhf = vj - .5 * vk
        e = float(np.einsum('ij,ji', h1e, dm) + .5 * np.einsum('ij,ji', vhf, dm) + mol.energy_nuc())
        f = h1e + vhf
        # orbital gradient 2 (1 - P S)^T... in the AO basis: g = F D S - S D F (the CDIIS error, scf/diis.py:89-96), and the
        # idempotency / electron count of the density in the oracle's own overlap
        comm = f.dot(dm).dot(s1e)
        comm = comm - comm.T
        res.update({'conv_e_rhf_functional': e, 'conv_fds_sdf_norm': float(np.linalg.norm(comm)),
                    'conv_nelec': float(np.einsum('ij,ji', dm, s1e)),
                    'conv_idempotency': float(np.linalg.norm(dm.dot(s1e).dot(dm) - 2 * dm)),
                    'conv_note': "oracle's DF-RHF energy functional and |FDS - SDF| at the density of --orbitals"})
        log('E_oracle[D_conv] = %.12f  (source reported %.12f)  |FDS-SDF| = %.3e' % (
            e, res['conv_e_tot_of_the_orbital_source'], res['conv_fds_sdf_norm']))
    with open(out_json, 'w') as fjs:
        json.dump(res, fjs, indent=1)
    log('%s: fp(vj) %.12f fp(vk) %.12f' % (name, res[key + 'vj_fp'], res[key + 'vk_fp']))
log('written', out_json)
