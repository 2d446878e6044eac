import sys, time, gc
sys.path.insert(0, '/root/repo')
import numpy as np
from pyscf_amd import gto, lib
from pyscf_amd.data import clusters
from pyscf_amd.df import native
mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
nao, nocc = mol.nao, mol.nelectron // 2
import os
for kv in filter(None, os.environ.get('PAMD_TUNE', '').split(',')):
    k_, v_ = kv.split('=')
    native.load().PAMD_set_tuning(k_.encode(), int(v_))
obj = native.NativeDF(mol, devices=[0]).build()
c = np.linalg.qr(np.random.RandomState(1).rand(nao, nao))[0]
occ = np.zeros(nao); occ[:nocc] = 2
dm = lib.tag_array((c * occ).dot(c.T), mo_coeff=c, mo_occ=occ, dm_from_orbitals=True)
obj.get_jk(dm, hermi=1)
ts, cs = [], []
for i in range(int(os.environ.get('NCALL', '14'))):
    t0 = time.perf_counter(); vj, vk = obj.get_jk(dm, hermi=1); ts.append((time.perf_counter() - t0) * 1e3)
    tm = obj.last_timing(); cs.append((round(tm['compute_ms'][0], 1), round(tm['e2_ms'][0], 1), round(tm['syrk_ms'][0], 1)))
print('python ms', [round(t, 1) for t in ts])
print('C compute / e2 / syrk', cs)
