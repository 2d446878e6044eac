import faulthandler, sys, time, os
faulthandler.enable()
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
def log(*a):
    print('[%.1fs]' % (time.time() - t0), *a, flush=True)
import numpy as np, torch
from pyscf_amd import lib, df
from oracle import ref
nao, naux, nocc = 257, 96, 161
rng = np.random.default_rng(nao)
npair = nao * (nao + 1) // 2
cderi = rng.standard_normal((naux, npair)) / np.sqrt(nao)
c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
occ = np.zeros(nao); occ[:nocc] = 2
dm = (c * occ).dot(c.T)
obj = df.DF(None); obj._cderi = cderi; obj.build()
log('built')
vj, _ = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1, with_k=False)
torch.cuda.synchronize(); log('vj done')
_, vk = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1, with_j=False)
torch.cuda.synchronize(); log('vk mo done')
full = ref.unpack_tril(cderi)
log('unpacked')
tmp = np.einsum('Lpq,qr->Lpr', full, dm, optimize=True)
vk0 = np.einsum('Lpr,Lqr->pq', tmp, full, optimize=True)
log('cpu vk0', abs(vk - vk0).max())
dms = rng.standard_normal((3, nao, nao))
vj, _ = obj.get_jk(dms, hermi=0, with_k=False)
torch.cuda.synchronize(); log('vj3 done')
_, vk = obj.get_jk(dms, hermi=0, with_j=False)
torch.cuda.synchronize(); log('vk general done')
vk0 = np.einsum('Lij,sjk,Lkl->sil', full, dms, full, optimize=True)
log('cpu general', abs(vk - vk0).max())
