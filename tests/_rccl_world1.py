"""Worker of tests/test_gpu_rccl.py: ONE rank on ONE GPU with the RCCL backend (`init_process_group('nccl', world_size=1,
device_id=...)`, exactly bench.py's call) and PAMD_FORCE_COLLECTIVE=1, so that every collective of the N > 1 path - the packed
f64 [J~ || K] all-reduce (df_jk._allreduce_jk_packed), the per-result all-reduces of the general branch, NumInt's vmat / nelec /
exc all-reduce, DF.loop()'s gather, barrier - really runs through RCCL.  A sum over one rank must change nothing: results are
compared with the same calls made without collectives.  Prints RCCL_WORLD1_OK."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    from pyscf_amd import gto, df, dft, lib
    from pyscf_amd.lib import comm
    from pyscf_amd.data import clusters
    from pyscf_amd.df import df_jk
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')
    obj = df.DF(mol).build()
    nao, nocc = mol.nao, mol.nelectron // 2
    rng = np.random.default_rng(4)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = lib.tag_array((c[:, :nocc] * 2).dot(c[:, :nocc].T), mo_coeff=c, mo_occ=occ)
    dms = rng.standard_normal((2, nao, nao))
    grids = dft.Grids(mol)
    grids.level = 1
    grids.build()
    ni = dft.NumInt()

    def run():
        vj, vk = obj.get_jk(dm, hermi=1)                    # MO branch -> packed all-reduce
        gj, gk = obj.get_jk(dms, hermi=0)                   # general branch -> one all-reduce per result
        n, e, v = ni.nr_rks(mol, grids, 'b3lyp', dm)
        full = np.vstack(list(obj.loop(50)))
        return vj, vk, gj, gk, np.array([n, e]), v, full

    comm.force(False)
    ref = run()
    comm.force(True)
    timer = comm.CommTimer()
    comm.set_timer(timer)
    assert comm.active(1) and comm.active(obj.world_size)
    got = run()
    ms, nbytes = timer.total_ms()
    comm.set_timer(None)
    assert len(timer.records) >= 5, len(timer.records)       # packed J/K, J + K general, vmat + acc, loop blocks
    for a, b in zip(ref, got):
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(a).max()), np.abs(a - b).max()
    # the collective itself in f64 on a device buffer of the size bench.py moves at config 3 (2 x nao_pair doubles)
    buf = torch.arange(2 * 1723296, dtype=torch.float64, device=dev)
    chk = buf.clone()
    dist.all_reduce(buf)
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(buf, chk)
    print('RCCL_WORLD1_OK collectives=%d bytes=%d ms=%.3f' % (len(timer.records), nbytes, ms), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
