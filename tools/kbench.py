"""Kernel micro-benchmark on a synthetic (random) cderi of a given size - no integral build.
    python tools/kbench.py [--nao 1856 --naux 4448 --nocc 160 --steps 3]
Prints per-kernel ms and rates (HIP events on the launch stream)."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_amd import df
from pyscf_amd.df import df_jk

ap = argparse.ArgumentParser()
ap.add_argument('--nao', type=int, default=1856)
ap.add_argument('--naux', type=int, default=4448)
ap.add_argument('--nocc', type=int, default=160)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--tag', default='')
ap.add_argument('--tune', default='', help='comma list key=value for PAMD_set_tuning')
ap.add_argument('--no-overlap', action='store_true')
ap.add_argument('--no-split', action='store_true')
ap.add_argument('--no-square', action='store_true')
ap.add_argument('--e2-pipeline', type=int, default=0)
ap.add_argument('--j2-policy', default='overlap', choices=['auto', 'overlap', 'serial', 'fused'])
ap.add_argument('--block-gb', type=float, default=0)
ap.add_argument('--no-fuse', action='store_true')
ap.add_argument('--ksplit', type=int, default=0)
ap.add_argument('--syrk-reserve', type=int, default=0, help='balanced re-tiled SYRK beside the co-running J pass, this many workgroup slots left free')
ap.add_argument('--syrk-flags', type=int, default=-1, help='4: balanced k split, 8: re-tiled triangle (df_jk.syrk_plan)')
ap.add_argument('--layout', default='packed', choices=['packed', 'square'], help="'square': the square rows as the only copy (r06), generated directly")
ap.add_argument('--sq-pad', type=int, default=-1, help='doubles added to the aux-row stride of the square layout (A/B of DF.SQ_STRIDE_PAD)')
ap.add_argument('--sq-contiguous', action='store_true', help='square layout WITHOUT the padded aux-row stride (A/B of DF.SQ_STRIDE_PAD)')
ap.add_argument('--side-priority', type=int, default=0, help='queue priority of the J side stream (1: lowest, 0: default)')
ap.add_argument('--no-j', action='store_true', help='K only (as the K_LR / response calls do)')
a = ap.parse_args()
dev = torch.device('cuda', 0)
npair = a.nao * (a.nao + 1) // 2
obj = df.DF(None)
g = torch.Generator(device=dev); g.manual_seed(1)
if a.sq_pad >= 0:
    df.DF.SQ_STRIDE_PAD = a.sq_pad
if a.layout == 'square':
    # symmetric random rows straight into the square layout (the packed tensor never exists: taxol shape = 225 GB of square rows)
    rows = (a.nao + 15) // 16 * 16
    sq = df.DF.alloc_square(a.naux, rows, dev) if not a.sq_contiguous else \
        torch.zeros(a.naux * rows * rows + 256, dtype=torch.float64, device=dev)[:a.naux * rows * rows].view(a.naux, rows, rows)
    for b0 in range(0, a.naux, 64):
        blk = sq[b0:b0 + 64]
        t = torch.empty((blk.shape[0], a.nao, a.nao), dtype=torch.float64, device=dev).normal_(generator=g)
        t = torch.tril(t)
        t = t + t.transpose(1, 2) - torch.diag_embed(torch.diagonal(t, dim1=1, dim2=2))
        blk[:, :a.nao, :a.nao] = t * (1.0 / np.sqrt(a.nao))
        del t
    obj._cderi_sq, obj._sq_nao, obj._layout = sq, a.nao, 'square'
else:
    obj._cderi_dev = torch.empty((a.naux, npair), dtype=torch.float64, device=dev)
    for b0 in range(0, a.naux, 256):
        obj._cderi_dev[b0:b0 + 256].normal_(generator=g)
    obj._cderi_dev.mul_(1.0 / np.sqrt(a.nao))
obj._naux = a.naux
obj.overlap_jk = not a.no_overlap
obj.overlap_split = not a.no_split
if a.no_square: obj.k_square = False
if a.e2_pipeline: obj.k_e2_pipeline = a.e2_pipeline
obj.j2_policy = a.j2_policy
if a.block_gb: obj.k_block_bytes = int(a.block_gb * (1 << 30))
if a.no_fuse: obj.fuse_j_pass1 = False
if a.ksplit: obj.k_nsplit = a.ksplit
obj.k_syrk_reserve = a.syrk_reserve
obj.k_side_priority = a.side_priority
obj.k_syrk_flags = None if a.syrk_flags < 0 else a.syrk_flags
import ctypes
from pyscf_amd import lib as _L
for kv in filter(None, a.tune.split(',')):
    k, v = kv.split('=')
    _L.check(_L.load_library().PAMD_set_tuning(k.encode(), int(v)))
rng = np.random.default_rng(1)
c = np.linalg.qr(rng.standard_normal((a.nao, a.nocc)))[0] * np.sqrt(2.0)
dm = c.dot(c.T)
dms = torch.from_numpy(dm[None]).to(dev)
orb = [df_jk.pad_orbitals(c, dev)]
df_jk.get_jk_device(obj, dms, orb, not a.no_j, True)
torch.cuda.synchronize()
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    vj, vk = df_jk.get_jk_device(obj, dms, orb, not a.no_j, True)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / a.steps * 1e3
obj.kernel_timer = df_jk.KernelTimer()
for _ in range(a.steps):
    vj, vk = df_jk.get_jk_device(obj, dms, orb, not a.no_j, True)
s = obj.kernel_timer.summary()
fl = 2.0 * a.naux * a.nao * a.nao * a.nocc
by = 8.0 * a.naux * npair
out = {'tag': a.tag, 'wall_ms': round(wall, 2), 'sum_kernel_ms': round(sum(t for t, _ in s.values()) / a.steps, 2)}
for k, (t, n) in s.items():
    ms = t / a.steps
    out[k] = round(ms, 3)
    if k in ('e2_symm', 'dgemm_tn'):
        out[k + '_TF'] = round(fl / ms / 1e9, 1)
    if k.startswith('vj_pass'):
        out[k + '_GBs'] = round(by / ms / 1e6, 0)
# fingerprints of the full-size results: runs with different tuning keys on the same synthetic tensor must agree
_w = torch.cos(torch.arange(a.nao * a.nao, dtype=torch.float64, device=dev)).view(a.nao, a.nao)
out['fp_vk'] = float((vk[0] * _w).sum())
if vj is not None:
    out['fp_vj'] = float((vj[0] * torch.cos(torch.arange(vj.shape[1], dtype=torch.float64, device=dev))).sum())
# cheap correctness probe on a few entries (fp64 reference on device for 32 aux rows)
sub = obj.packed_rows(0, 32).clone()
idx = torch.tril_indices(a.nao, a.nao, device=dev)
full = torch.zeros((32, a.nao, a.nao), dtype=torch.float64, device=dev)
full[:, idx[0], idx[1]] = sub
full = full + full.transpose(1, 2) - torch.diag_embed(torch.diagonal(full, dim1=1, dim2=2))
o2 = df.DF(None); o2._cderi_dev = sub
vj2, vk2 = df_jk.get_jk_device(o2, dms, orb)
cd = torch.from_numpy(c).to(dev)
xx = torch.matmul(full, cd)                         # [L][p][i]
vk_ref = torch.einsum('Lpi,Lqi->pq', xx, xx)
rho = torch.einsum('Lpq,pq->L', full, dms[0])
vj_ref = torch.einsum('L,Lpq->pq', rho, full)
vj_full = torch.zeros((a.nao, a.nao), dtype=torch.float64, device=dev)
vj_full[idx[0], idx[1]] = vj2[0]
out['err_vk'] = float((vk2[0] - vk_ref).abs().max() / vk_ref.abs().max())
out['err_vj'] = float((vj_full.tril() - vj_ref.tril()).abs().max() / vj_ref.abs().max())
if a.j2_policy == 'fused':
    # the fused kernel against the in-line schedule on the full tensor
    obj.kernel_timer = None
    obj.j2_policy = 'serial'
    vj_s, vk_s = df_jk.get_jk_device(obj, dms, orb, not a.no_j, True)
    out['fused_vs_serial_vj'] = float((vj - vj_s).abs().max() / vj_s.abs().max())
    out['fused_vs_serial_vk'] = float((vk - vk_s).abs().max() / vk_s.abs().max())
print(json.dumps(out))
