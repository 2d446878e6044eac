"""One DF.build of (H2O)_32 cc-pVTZ (int3c2e family + cderi_solve) - the target of the build-path counter passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_amd import gto, df
from pyscf_amd.data import clusters
mol = gto.M(atom=clusters.water_cluster(int(os.environ.get('NWATER', '32'))), basis='cc-pvtz')
obj = df.DF(mol, auxbasis='cc-pvtz-jkfit')
obj.k_square = False
t0 = time.perf_counter()
obj.build()
torch.cuda.synchronize()
print('build %.2f s naux %d' % (time.perf_counter() - t0, obj.get_naoaux()))
