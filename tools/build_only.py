"""DF.build of (H2O)_n cc-pVTZ (int3c2e family + cderi_solve) - the target of the build-path counter passes; with --time the
HIP-event time of every build phase and the roofline of the solve (naux^2 nao_pair flops on the triangular factor).
    python tools/build_only.py [--time] [--layout packed|square|auto]      (NWATER=32; PAMD_SOLVE_V2=0: the r01 solve kernel)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_amd import gto, df
from pyscf_amd.data import clusters
mol = gto.M(atom=clusters.water_cluster(int(os.environ.get('NWATER', '32'))), basis='cc-pvtz')
layout = sys.argv[sys.argv.index('--layout') + 1] if '--layout' in sys.argv else 'packed'
times = []
for rep in range(3 if '--time' in sys.argv else 1):
    obj = df.DF(mol, auxbasis='cc-pvtz-jkfit')
    obj.layout = layout
    if layout == 'packed':
        obj.k_square = False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obj.build()
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    naux, npair = obj.get_naoaux(), obj.tensor_shape()[1]
    lay = obj._layout
    obj.reset()
    del obj
    torch.cuda.empty_cache()
print('build %.2f s naux %d' % (times[-1], naux))
if '--time' in sys.argv:
    print(json.dumps({'layout': lay, 'solve_v2': os.environ.get('PAMD_SOLVE_V2', '1') != '0', 'build_s': [round(t, 3) for t in times],
                      'naux': naux, 'nao_pair': npair, 'solve_flops': float(naux) * naux * npair}))
