import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from pyscf_amd import gto, dft
from pyscf_amd.data import clusters
mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz', verbose=0)
mf = dft.RKS(mol, xc='b3lyp').density_fit()
mf.max_cycle = 3
pr = cProfile.Profile(); pr.enable()
mf.kernel()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:6000])
