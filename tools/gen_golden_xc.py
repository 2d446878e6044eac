"""Oracle-only XC golden at sizes whose dense AO matrix does not fit the CPU box (taxol def2-TZVP: 1.37 M grid points x 2228 AOs x 4
components = 98 GB): oracle/ref_dft.nr_rks - numpy Becke grid, numpy AO values, sympy-differentiated functional - evaluated grid block
by grid block (nelec, exc and vmat are sums over grid points, numint.py:1116-1157) at one or more densities given by their occupied
orbitals.  No product code on the path (pyscf_amd is used for the molecule tables and the atomic radial / Lebedev tables only, like
every other oracle golden).

    python tools/gen_golden_xc.py --molecule taxol --xc b3lyp --orbitals rhf=gpurun_out/taxol_rhf_orbitals.npz \
        b3lyp=gpurun_out/taxol_dump/taxol_b3lyp_orbitals.npz

Adds to tests/golden/<tag>_oracle.json, per density name N:  xc_N_nelec, xc_N_exc, xc_N_vxc_fp, xc_N_vxc_norm, xc_N_vxc_sample (4096
seeded entries), xc_N_tr_d_vxc; and - when the J/K traces of the same density are already in a golden file (--jk-json, keys
conv_tr_d_vj / conv_tr_d_vk / conv_e_rhf_functional written by tools/gen_golden_streaming.py) - the oracle's DF-RKS energy functional
E[D] = E_RHF[D] + (1 - hyb)/4 Tr(D K) + E_xc[D]  (pyscf/dft/rks.py:76-131, energy_elec :147-181)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_dft, golden_util      # noqa: E402
from pyscf_amd import gto                     # noqa: E402  (host-only tables)
from pyscf_amd.data import clusters           # noqa: E402
from pyscf_amd.dft import libxc               # noqa: E402  (host-only: xc string -> component weights)

ap = argparse.ArgumentParser()
ap.add_argument('--molecule', default='taxol', choices=['taxol', 'water'])
ap.add_argument('--nwater', type=int, default=2)
ap.add_argument('--basis', default=None)
ap.add_argument('--xc', default='b3lyp')
ap.add_argument('--level', type=int, default=3)
ap.add_argument('--orbitals', nargs='+', required=True, help='name=file.npz (orbo = C_occ sqrt(occ)); the name "syn<r>" = seeded')
ap.add_argument('--jk-json', nargs='*', default=[], help='name=golden.json holding conv_tr_d_vj / conv_tr_d_vk / conv_e_rhf_functional of that density')
ap.add_argument('--block', type=int, default=16384)
ap.add_argument('--nsample', type=int, default=4096)
ap.add_argument('--tag', default='')
ap.add_argument('--combine-only', action='store_true', help='no grid work: only (re)compute the energy functionals from stored exc + --jk-json')
a = ap.parse_args()
t00 = time.time()


def log(*args):
    print('[%7.1fs]' % (time.time() - t00), *args, flush=True)


if a.basis is None:
    a.basis = 'def2-tzvp' if a.molecule == 'taxol' else 'cc-pvdz'
atoms = clusters.taxol() if a.molecule == 'taxol' else clusters.water_cluster(a.nwater)
mol = gto.M(atom=atoms, basis=a.basis)
nao = mol.nao
tag = a.tag or ('taxol_%s' % a.basis.replace('-', '') if a.molecule == 'taxol' else 'h2o%d_%s' % (a.nwater, a.basis.replace('-', '')))
out_json = os.path.join(ROOT, 'tests', 'golden', tag + '_oracle.json')
res = json.load(open(out_json)) if os.path.exists(out_json) else {}
hyb, fac = libxc.parse_xc(a.xc)
gga = 1 if libxc.xc_type(a.xc) == 'GGA' else 0
if a.combine_only:
    for spec in a.jk_json:
        name, path = spec.split('=', 1)
        g = json.load(open(path))
        k = 'xc_%s_' % name
        e_rks = g['conv_e_rhf_functional'] + 0.25 * (1.0 - hyb) * g['conv_tr_d_vk'] + res[k + 'exc']
        res.update({k + 'e_rks_functional': e_rks, k + 'e_rks_note': "oracle's DF-RKS energy functional at this density: its DF-RHF functional "
                    "(%s: conv_e_rhf_functional) + (1 - hyb)/4 Tr(D K) + E_xc, hyb = %g" % (os.path.basename(path), hyb),
                    k + 'e_tot_of_the_orbital_source': g.get('conv_e_tot_of_the_orbital_source')})
        print('%s: E_RKS[D] (oracle functional) = %.12f   orbital source reported %.12f' % (name, e_rks, g.get('conv_e_tot_of_the_orbital_source', 0.0)))
    with open(out_json, 'w') as f:
        json.dump(res, f, indent=1)
    sys.exit(0)
sets = []
for spec in a.orbitals:
    name, path = spec.split('=', 1) if '=' in spec else (spec, '')
    if name.startswith('syn'):
        c = golden_util.synthetic_orbitals(nao, int(name[3:] or 32)) * np.sqrt(2.0)
        desc = 'D = 2 C C^T, C = oracle.golden_util.synthetic_orbitals(nao, %d)' % c.shape[1]
    else:
        c = np.ascontiguousarray(np.load(path)['orbo'])
        desc = 'occupied orbitals of %s' % os.path.basename(path)
    assert c.shape[0] == nao
    sets.append((name, c, desc))
log('nao', nao, 'xc', a.xc, 'hyb', hyb, 'gga', gga, 'densities', [(n, c.shape[1]) for n, c, _ in sets])
coords, weights = ref_dft.build_grids(mol, level=a.level)
ng = len(weights)
log('oracle grid (level %d, Treutler radial, NWChem pruning, Becke partition): %d points' % (a.level, ng))
dms = [c.dot(c.T) for _, c, _ in sets]
acc = [[0.0, 0.0, np.zeros((nao, nao))] for _ in sets]
for g0 in range(0, ng, a.block):
    g1 = min(g0 + a.block, ng)
    t = time.time()
    for s, dm in enumerate(dms):
        # (the AO block is re-evaluated per density: ref_dft.nr_rks is called as it stands, no second code path in the oracle)
        n, e, v = ref_dft.nr_rks(mol, coords[g0:g1], weights[g0:g1], fac, gga, dm)
        acc[s][0] += float(n)
        acc[s][1] += float(e)
        acc[s][2] += v
    if (g0 // a.block) % 8 == 0:
        log('grid block [%d, %d) %.1f s' % (g0, g1, time.time() - t))
ri, ci = golden_util.sample_positions(nao, a.nsample)
jk = dict(s.split('=', 1) for s in a.jk_json)
res.update({'xc_code': a.xc, 'xc_grid': 'level %d, %d points (oracle/ref_dft.build_grids defaults = gen_grid.py:565-576)' % (a.level, ng),
            'xc_ngrids': ng, 'xc_generator': 'tools/gen_golden_xc.py (CPU oracle only: oracle/ref_dft.py numpy AO values + sympy functional)',
            'xc_sample_seed': 11})
for s, (name, c, desc) in enumerate(sets):
    nelec, exc, vxc = acc[s]
    k = 'xc_%s_' % name
    res.update({k + 'density': desc, k + 'nelec': nelec, k + 'exc': exc, k + 'vxc_fp': golden_util.fp(vxc),
                k + 'vxc_norm': float(np.linalg.norm(vxc)), k + 'vxc_absmax': float(abs(vxc).max()),
                k + 'tr_d_vxc': float(np.einsum('ij,ji', dms[s], vxc)), k + 'vxc_sample': [float(x) for x in vxc[ri, ci]]})
    log('%s: nelec %.10f exc %.12f fp(vxc) %.12f' % (name, nelec, exc, res[k + 'vxc_fp']))
    if name in jk:
        g = json.load(open(jk[name]))
        e_rks = g['conv_e_rhf_functional'] + 0.25 * (1.0 - hyb) * g['conv_tr_d_vk'] + exc
        res.update({k + 'e_rks_functional': e_rks, k + 'e_rks_note': "oracle's DF-RKS energy functional at this density: its DF-RHF functional "
                    "(%s: conv_e_rhf_functional) + (1 - hyb)/4 Tr(D K) + E_xc, hyb = %g" % (os.path.basename(jk[name]), hyb)})
        log('%s: E_RKS[D] (oracle functional) = %.12f' % (name, e_rks))
with open(out_json, 'w') as f:
    json.dump(res, f, indent=1)
log('written', out_json)
