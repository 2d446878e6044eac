"""The host-array C ABI (PAMD_df_create / PAMD_df_get_jk) at BASELINE config 3, numpy only (no torch in the process):
ms per J/K call with the schedule of the second J pass chosen by the handle, and with each schedule forced.
    python tools/native_bench.py [--nwater 32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_amd import gto, lib
from pyscf_amd.data import clusters
from pyscf_amd.df import native
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default='cc-pvtz')
a = ap.parse_args()
assert 'torch' not in sys.modules
mol = gto.M(atom=clusters.water_cluster(a.nwater), basis=a.basis)
nao, nocc = mol.nao, mol.nelectron // 2
t0 = time.perf_counter()
obj = native.NativeDF(mol).build()
print('PAMD_df_create: %.1f s (nao %d, naux %d)' % (time.perf_counter() - t0, nao, obj.get_naoaux()), flush=True)
c = np.linalg.qr(np.random.RandomState(1).rand(nao, nao))[0]
occ = np.zeros(nao)
occ[:nocc] = 2
dm = lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ, dm_from_orbitals=True)
ref = None
for sched in ('auto', 'overlap', 'serial', 'fused'):
    if sched == 'auto':
        os.environ.pop('PAMD_DF_J2', None)
    else:
        os.environ['PAMD_DF_J2'] = sched
    t0 = time.perf_counter()
    vj, vk = obj.get_jk(dm, hermi=1)
    first = time.perf_counter() - t0
    if sched == 'auto':
        # r06: the handle times its schedules on the caller's own calls (one candidate per call, two samples each): 7 more calls settle it
        settle = []
        for _ in range(7):
            t1 = time.perf_counter()
            obj.get_jk(dm, hermi=1)
            settle.append(round((time.perf_counter() - t1) * 1e3, 1))
        print('auto     calls 2-8 (overlap / serial / fused in turn): %s ms' % settle, flush=True)
    t0 = time.perf_counter()
    for _ in range(3):
        vj, vk = obj.get_jk(dm, hermi=1)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    if ref is None:
        ref = (vj, vk)
    print('%-8s first call %.2f s, then %.1f ms per call (host arrays in / out); |dJ| %.1e |dK| %.1e vs the first schedule'
          % (sched, first, ms, np.abs(vj - ref[0]).max(), np.abs(vk - ref[1]).max()), flush=True)
assert 'torch' not in sys.modules
