import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from pyscf_amd import gto, dft
from pyscf_amd.data import clusters
mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
grids = dft.Grids(mol).build()
ni = dft.NumInt()
dev = torch.device('cuda', 0)
nao = mol.nao; ldao = (nao + 15) // 16 * 16
coords = torch.from_numpy(grids.coords).to(dev)
for thr in (1e-13, 1e-10, 1e-8):
    tot = {}
    for g0 in (0, 300000, 600000, 900000):
        blk = 65536
        ao = torch.zeros((4, blk, ldao), dtype=torch.float64, device=dev)
        ni.eval_ao_block(mol, coords, g0, blk, 1, ao, blk, ldao)
        a = ao.abs().amax(0)                         # max over comps
        m16 = a.view(blk // 16, 16, ldao // 16, 16).amax((1, 3)) > thr        # [g16][mu16]
        pad = (-m16.shape[1]) % 8
        m16p = torch.nn.functional.pad(m16, (0, pad))
        mA = m16p.view(m16.shape[0], -1, 8).any(-1)                              # [kt][ct128]
        m2 = m16.view(m16.shape[0] // 8, 8, -1).any(1)                           # [g128][mu16]
        f = mA.float()
        pair = (f.t() @ f)                                                       # [tm][tn] counts of active k-tiles
        dens_pair = float(pair.sum() / (mA.shape[0] * mA.shape[1] ** 2))
        tot.setdefault('m16', []).append(float(m16.float().mean()))
        tot.setdefault('maskA(16x128)', []).append(float(mA.float().mean()))
        tot.setdefault('mask2(128x16)', []).append(float(m2.float().mean()))
        tot.setdefault('vmat pair density', []).append(dens_pair)
    print(thr, {k: round(float(np.mean(v)), 3) for k, v in tot.items()}, flush=True)
