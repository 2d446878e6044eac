"""Step-by-step GPU probe with flushed progress lines (debug aid for gpurun)."""
import faulthandler, sys, time, os
faulthandler.enable()
faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
def log(*a):
    print('[%.1fs]' % (time.time() - t0), *a, flush=True)
log('start')
import numpy as np
import torch
log('torch', torch.__version__, torch.cuda.is_available())
x = torch.zeros(10, device='cuda'); torch.cuda.synchronize()
log('torch cuda ok', torch.cuda.get_device_name(0))
from pyscf_amd import lib
L = lib.load_library()
log('lib loaded; hip libs mapped:', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l}))
log('device count', L.PAMD_device_count())
import ctypes
from pyscf_amd.df import df_jk
nao = 24; npair = nao*(nao+1)//2
dm = torch.rand(1, nao, nao, dtype=torch.float64, device='cuda')
tril = torch.zeros(1, npair, dtype=torch.float64, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
log('stream', st)
rc = L.PAMD_pack_dm_tril(ctypes.c_void_p(dm.data_ptr()), 1, nao, ctypes.c_void_p(tril.data_ptr()), st)
log('pack launched rc', rc)
torch.cuda.synchronize()
log('pack done', float(tril.sum()), float((dm[0]+dm[0].T).tril().sum() - dm[0].diagonal().sum()))
from oracle import ref
from pyscf_amd import gto, df
H2O = 'O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587'
mol = gto.M(atom=H2O, basis='cc-pvdz'); aux = gto.M(atom=H2O, basis='weigend')
cderi = ref.cholesky_eri(mol, aux)
log('oracle cderi', cderi.shape)
obj = df.DF(mol); obj._cderi = cderi; obj.build()
np.random.seed(1); dms = np.random.random((2, nao, nao))
vj, _ = obj.get_jk(dms, hermi=0, with_k=False)
log('vj fp', lib.fp(vj), -194.15910890730066)
_, vk = obj.get_jk(dms, hermi=0, with_j=False)
log('vk fp', lib.fp(vk), -46.365071587653517)
c = np.linalg.qr(np.random.random((nao, nao)))[0]; occ = np.zeros(nao); occ[:5] = 2
dmo = (c*occ).dot(c.T)
vj, vk = obj.get_jk(lib.tag_array(dmo, mo_coeff=c, mo_occ=occ), hermi=1)
vj0, vk0 = ref.get_jk(cderi, dmo, 1, mo_coeff=c, mo_occ=occ)
log('mo branch err', abs(vj-vj0).max(), abs(vk-vk0).max())
from pyscf_amd.df import incore
j3c = incore.aux_e2_gpu(mol, aux, torch.device('cuda', 0)).cpu().numpy()
log('int3c2e fp', ref.fp(j3c.T), 12.407403711205063, 'err', abs(j3c - ref.pack_tril(ref.int3c2e(mol, aux))).max())
cd = incore.cholesky_eri_gpu(mol, aux, torch.device('cuda', 0)).cpu().numpy()
log('cderi err', abs(cd - cderi).max())
from pyscf_amd import scf
mf = scf.RHF(mol).density_fit(auxbasis='weigend'); mf.conv_tol = 1e-10; mf.verbose = 4
e = mf.kernel()
log('E', e, -76.025936299702536, e + 76.025936299702536)
