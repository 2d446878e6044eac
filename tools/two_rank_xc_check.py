"""Run under torch.distributed.run with 2 ranks (gloo, one GPU): nr_rks at (H2O)_n with the grid tiles dealt over the two
ranks against the same call on all tiles (world override), on rank 0."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
from pyscf_amd import gto, dft, lib
from pyscf_amd.data import clusters
from pyscf_amd.scf import hf
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mol = gto.M(atom=clusters.water_cluster(nw), basis='cc-pvtz')
nao, nocc = mol.nao, mol.nelectron // 2
grids = dft.Grids(mol).build()
s1e = hf.int1e_gpu(mol, torch.device('cuda', 0))[0]
rng = np.random.RandomState(1)
x = rng.random_sample((nao, nao))
w, v = np.linalg.eigh(x.T.dot(s1e).dot(x))
c = x.dot(v / np.sqrt(w)).dot(v.T)
occ = np.zeros(nao); occ[:nocc] = 2
dm = lib.tag_array((c[:, :nocc] * 2).dot(c[:, :nocc].T), mo_coeff=c, mo_occ=occ)
ni = dft.NumInt()
n2, e2, v2 = ni.nr_rks(mol, grids, 'b3lyp', dm)
one = dft.NumInt()
one._world_override = (0, 1)
n1, e1, v1 = one.nr_rks(mol, grids, 'b3lyp', dm)
if rank == 0:
    print(json.dumps({'nelec_2rank': n2, 'nelec_1rank': n1, 'd_nelec': n2 - n1, 'd_exc': e2 - e1, 'd_vmat': float(abs(v2 - v1).max()),
                      'vmat_absmax': float(abs(v1).max())}))
dist.barrier()
dist.destroy_process_group()
