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

# ---- r03: does an HBM-bound stream beside the matrix pipe cost matrix throughput?  (power limit)
def _live(iters):
    lib.PAMD_mfma_f64_live(ctypes.c_void_p(out.data_ptr()), 512, iters, st)


def _timed(fn):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


iters = 90000
fl = 512 * 4 * iters * 20 * 2048.0
ms_alone = _timed(lambda: _live(iters))
buf = torch.empty(4 << 30, dtype=torch.float64, device='cuda')          # 32 GiB
buf.normal_()
side = torch.cuda.Stream()


def _stream_pass(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            buf.sum()


ms_stream = _timed(lambda: (_stream_pass(8), side.synchronize()))        # 8 x 32 GiB alone
gbs_alone = 8 * buf.numel() * 8 / ms_stream / 1e6


def _both():
    ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
    _stream_pass(8)
    _live(iters)
    torch.cuda.current_stream().wait_stream(side)


ms_both = _timed(_both)
res2 = {'mfma_live_alone_TFLOPs': round(fl / ms_alone / 1e9, 2), 'mfma_live_alone_ms': round(ms_alone, 2),
        'hbm_stream_alone_GBs': round(gbs_alone, 0), 'hbm_stream_alone_ms': round(ms_stream, 2),
        'both_concurrently_ms': round(ms_both, 2), 'sum_of_alone_ms': round(ms_alone + ms_stream, 2),
        'note': 'both_concurrently_ms close to the SUM of the two: the HBM stream does not hide behind the matrix pipe'}
print(json.dumps(res2))
