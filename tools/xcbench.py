"""nr_rks micro-benchmark: (H2O)_n cc-pVTZ, level-3 grid, B3LYP, per-kernel HIP-event timings.
    python tools/xcbench.py [--nwater 32 --xc b3lyp --steps 2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import gto, dft, lib
from pyscf_amd.data import clusters
from pyscf_amd.df import df_jk
from pyscf_amd.scf import hf

ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
ap.add_argument('--xc', default='b3lyp')
ap.add_argument('--steps', type=int, default=2)
ap.add_argument('--nsplit', type=int, default=0)
ap.add_argument('--dense', action='store_true')
ap.add_argument('--tile', type=int, default=0)
ap.add_argument('--cutoff', type=float, default=0)
ap.add_argument('--vmat-sym', type=int, default=1, help='1: PAMD_sub_vmat_sym (r04), 0: the r03 kernel')
ap.add_argument('--chunk', type=int, default=0, help='grid points per launch group')
ap.add_argument('--tune-xc', default='', help='comma list key=value for PAMD_set_tuning_xc')
a = ap.parse_args()
import ctypes
for kv in filter(None, a.tune_xc.split(',')):
    k, v = kv.split('=')
    lib.check(lib.load_library().PAMD_set_tuning_xc(k.encode(), int(v)))
dev = torch.device('cuda', 0)
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
nao, nocc = mol.nao, mol.nelectron // 2
t0 = time.perf_counter()
grids = dft.Grids(mol).build()
t_grid = time.perf_counter() - t0
s1e = hf.int1e_gpu(mol, dev)[0]
rng = np.random.RandomState(1)
x = rng.random_sample((nao, nao))
w, v = np.linalg.eigh(x.T.dot(s1e).dot(x))
c = x.dot(v / np.sqrt(w)).dot(v.T)
occ = np.zeros(nao); occ[:nocc] = 2
dm = lib.tag_array((c[:, :nocc] * 2).dot(c[:, :nocc].T), mo_coeff=c, mo_occ=occ)
ni = dft.NumInt()
if a.nsplit: ni.vmat_nsplit = a.nsplit
if a.dense: ni.sparse = False
if a.tile: ni.sparse_tile = a.tile
if a.cutoff: ni.sparse_cutoff = a.cutoff
ni.vmat_sym = bool(a.vmat_sym)
if a.chunk: ni.sparse_chunk_points = a.chunk
n, e, vm = ni.nr_rks(mol, grids, a.xc, dm)
torch.cuda.synchronize()
ni.kernel_timer = df_jk.KernelTimer()
t0 = time.perf_counter()
for _ in range(a.steps):
    n, e, vm = ni.nr_rks(mol, grids, a.xc, dm)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps
s = ni.kernel_timer.summary()
out = {'nsplit': a.nsplit, 'dense': a.dense, 'nao': nao, 'nocc': nocc, 'ngrids': int(grids.size), 'grid_build_s': round(t_grid, 2), 'nelec': float(n),
       'nelec_exact': mol.nelectron, 'exc': float(e), 'wall_ms_per_call': round(wall * 1e3, 1),
       'kernel_ms': {k: round(t / a.steps, 2) for k, (t, c_) in s.items()}}
ng = grids.size
out['TF'] = {'ao_dot_mo': round(4 * 2.0 * ng * nao * ((nocc + 15) // 16 * 16) / (s['ao_dot_mo'][0] / a.steps) / 1e9, 1),
             'ao_dot_aow': round(2.0 * ng * nao * nao / (s['ao_dot_aow'][0] / a.steps) / 1e9, 1)}
if ni.sparse:
    plan = ni.sparse_plan(mol, grids, True)
    out['plan'] = {'G': plan.G, 'tiles': plan.nloc, 'density': plan.density, 'density2': plan.density2,
                   'compact_GB': plan.ao_total * 8e-9, 'cached': plan.ao_c is not None}
    out['TF']['ao_dot_mo'] = round(out['TF']['ao_dot_mo'] * plan.density, 1)
    out['TF']['ao_dot_aow'] = round(out['TF']['ao_dot_aow'] * plan.density2, 1)
    # executed flops of the two MFMA products from the plan itself (16-column groups incl. padding): roofline of the XC leg
    ldh = plan.ld_host.astype(np.float64)
    npad = (nocc + 15) // 16 * 16
    fl_mo = 2.0 * plan.ncomp * plan.G * float(ldh.sum()) * npad
    fl_vm = 2.0 * plan.G * float((ldh ** 2).sum())
    # r06: how much of sub_vmat_sym's matrix-pipe time is lost to the uneven split of a piece between the two wave rows / columns
    # (7 groups = 4 + 3: the 3 x 3 wave idles while the 4 x 4 wave of the same workgroup multiplies): useful / (4 x slowest wave)
    def _balance(ldh):
        use = tot = 0.0
        hist = {}
        for g in (ldh / 16).astype(int):
            hist[int(g)] = hist.get(int(g), 0) + 1
        for g, cnt in hist.items():
            npc = (g + 7) // 8
            if npc == 0:
                continue
            base, rem = divmod(g, npc)
            pcs = [base + (1 if i < rem else 0) for i in range(npc)]
            for i in range(npc):
                for j in range(i + 1):
                    ph = 1 if i == j else 2
                    use += cnt * ph * pcs[i] * pcs[j]
                    tot += cnt * ph * 4 * ((pcs[i] + 1) // 2) * ((pcs[j] + 1) // 2)
        return round(use / tot, 4), dict(sorted(hist.items()))
    eff, hist = _balance(ldh)
    out['vmat_sym_wave_balance_model'] = eff
    out['ld_groups_hist'] = hist
    out['vmat_sym'] = bool(ni.vmat_sym)
    out['executed'] = {k: {'TF': round(f * 1e-12, 4), 'ms': round(s[k][0] / a.steps, 2), 'TFs': round(f / (s[k][0] / a.steps) * 1e-9, 1),
                           'frac_of_78.6': round(f / (s[k][0] / a.steps) * 1e-9 / 78.6, 3)}
                       for k, f in (('ao_dot_mo', fl_mo), ('ao_dot_aow', fl_vm))}
print(json.dumps(out))
