"""FP64 MFMA ceiling micro-benchmark (register-only v_mfma_f64_16x16x4_f64 streams)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyscf_amd import lib as L
lib = L.load_library()
out = torch.zeros(4, dtype=torch.float64, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
for nb in (256, 512, 2048):
    for nacc in (8, 20):
        for scale in (1.0, 0.0):
            iters = 160000 // nacc
            lib.PAMD_mfma_f64_peak(ctypes.c_void_p(out.data_ptr()), nb, 100, nacc, ctypes.c_double(scale), st)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.PAMD_mfma_f64_peak(ctypes.c_void_p(out.data_ptr()), nb, iters, nacc, ctypes.c_double(scale), st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            flops = nb * 4 * iters * nacc * 2048.0
            res['blocks=%d nacc=%d %s' % (nb, nacc, 'random' if scale else 'zeros')] = round(flops / ms / 1e9, 2)
print(json.dumps(res))
