"""Where a whole DF-RKS SCF spends its HOST time (VERDICT r05 Weak 10): cProfile of mf.kernel() at config 3, plus import / set-up clocks.
    python tools/prof_scf.py [--max-cycle 4] [--no-image]"""
import cProfile, pstats, sys, os, io, time
t_start = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t_torch = time.perf_counter()
from pyscf_amd import gto, dft, lib
from pyscf_amd.data import clusters
t_imp = time.perf_counter()
mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz', verbose=4)
t_mol = time.perf_counter()
lib.load_library()
torch.zeros(1, device='cuda')
torch.cuda.synchronize()
t_ctx = time.perf_counter()
mf = dft.RKS(mol, xc='b3lyp').density_fit()
if '--no-image' in sys.argv:
    mf.with_df.prefer_image = False        # square rows as the only copy (what taxol gets)
mf.max_cycle = int(sys.argv[sys.argv.index('--max-cycle') + 1]) if '--max-cycle' in sys.argv else 4
pr = cProfile.Profile(); pr.enable()
mf.kernel()
pr.disable()
t_end = time.perf_counter()
print('clocks: import torch %.2f s, import pyscf_amd %.2f s, gto.M %.2f s, library + HIP context %.2f s, kernel() %.2f s' % (
    t_torch - t_start, t_imp - t_torch, t_mol - t_imp, t_ctx - t_mol, t_end - t_ctx))
for key in ('cumulative', 'tottime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32); print(s.getvalue()[:5200])
