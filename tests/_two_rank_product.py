"""Worker of tests/test_gpu_bench_launch.py::test_two_rank_product_path: launched by torch.distributed.run with two ranks
sharing one GPU (gloo).  The aux-sharded PRODUCT path (tensor build per shard, J/K, nr_rks, save / reload of the shards)
against the CPU oracle; prints TWO_RANK_OK on rank 0."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import ref, ref_dft
    from pyscf_amd import gto, df, dft, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.dft import libxc
    mol = gto.M(atom=clusters.water_cluster(3), basis='cc-pvdz')
    obj = df.DF(mol).build()
    naux = obj.get_naoaux()
    l0, l1 = obj.shard_range(naux, rank, world)
    assert obj.world_size == world and obj._cderi_dev.shape[0] == l1 - l0 < naux
    cderi = ref.cholesky_eri(mol, df.make_auxmol(mol))
    assert np.abs(obj._cderi_dev.cpu().numpy() - cderi[l0:l1]).max() < 1e-9        # this rank's rows of the tensor
    nao, nocc = mol.nao, mol.nelectron // 2
    rng = np.random.default_rng(4)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c[:, :nocc] * 2).dot(c[:, :nocc].T)
    vj0, vk0 = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
    vj, vk = obj.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)       # MO branch, all-reduced over the shards
    assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9
    # DF.loop() on a sharded tensor is a collective that hands EVERY rank the FULL tensor (what a stock DF-MP2 / ao2mo consumer
    # iterates over, pyscf/df/df.py:214-242); loop(local=True) gives the rank's rows only
    got = np.vstack(list(obj.loop(37)))
    assert got.shape == cderi.shape and np.abs(got - cderi).max() < 1e-9
    assert np.vstack(list(obj.loop(37, local=True))).shape[0] == l1 - l0
    dms = rng.standard_normal((2, nao, nao))
    vj0, vk0 = ref.get_jk(cderi, dms, 0)
    vj, vk = obj.get_jk(dms, hermi=0)                                              # general-DM branch
    assert np.abs(vj - vj0).max() < 1e-9 and np.abs(vk - vk0).max() < 1e-9
    # shards written per rank, reloaded without re-sharding
    tmp = os.path.join(tempfile.gettempdir(), 'pamd_two_rank_cderi')
    out = obj.save(tmp, fmt='npy')
    assert out.endswith('.rank%dof%d.npz' % (rank, world))
    dist.barrier()
    obj2 = df.DF(mol)
    obj2._cderi = tmp
    obj2.build()
    assert obj2.get_naoaux() == naux and torch.equal(obj2._cderi_dev, obj._cderi_dev)
    vj2, vk2 = obj2.get_jk(dms, hermi=0)
    assert np.abs(vk2 - vk).max() < 1e-12
    # the reference's format: ONE HDF5 file, dataset 'j3c' (naux, nao_pair); the ranks write / read their row ranges
    from pyscf_amd.lib import hdf5
    if hdf5.available():
        h5 = tmp + '.h5'
        obj.save(h5)
        if rank == 0:
            with hdf5.File(h5) as f:
                d = f['j3c']
                assert d.shape == cderi.shape and np.abs(d.read_rows(0, naux) - cderi).max() < 1e-9
        dist.barrier()
        obj3 = df.DF(mol)
        obj3._cderi = h5
        obj3.build()
        assert obj3.get_naoaux() == naux and torch.equal(obj3._cderi_dev, obj._cderi_dev)
    # r05: a rank whose shard does not fit its device (cap) holds it out of core behind the C handle (part / nparts); the handle
    # returns the shard's PARTIAL J/K in host arrays and DF.get_jk all-reduces them over the ranks - here over gloo, for real
    oc = df.DF(mol)
    oc.outcore_device_bytes = 30 * (nao * (nao + 1) // 2) * 8
    oc.build()
    lay = oc.out_of_core()
    assert lay is not None and oc._cderi_dev is None and list(oc._native.shard_rows) == [l0, l1] and lay['rows_host'] > 0, lay
    vj3, vk3 = oc.get_jk(lib.tag_array(dm, mo_coeff=c, mo_occ=occ), hermi=1)
    vjr, vkr = ref.get_jk(cderi, dm, 1, mo_coeff=c, mo_occ=occ)
    assert np.abs(vj3 - vjr).max() < 1e-9 and np.abs(vk3 - vkr).max() < 1e-9
    vj3, vk3 = oc.get_jk(dms, hermi=0)
    assert np.abs(vj3 - vj0).max() < 1e-9 and np.abs(vk3 - vk0).max() < 1e-9
    assert np.abs(np.vstack(list(oc.loop(37, local=True))) - cderi[l0:l1]).max() < 1e-9
    oc.reset()
    # XC: grid tiles dealt round-robin, vmat / nelec / exc all-reduced
    grids = dft.Grids(mol)
    grids.level = 1
    grids.build()
    ni = dft.NumInt()
    ni.sparse_tile = 256
    n, e, v = ni.nr_rks(mol, grids, 'b3lyp', lib.tag_array(dm, mo_coeff=c, mo_occ=occ))
    hyb, fac = libxc.parse_xc('b3lyp')
    n0, e0, v0 = ref_dft.nr_rks(mol, grids.coords, grids.weights, fac, True, dm)
    assert abs(n - n0) < 1e-9 and abs(e - e0) < 1e-9 and np.abs(v - v0).max() < 1e-9
    assert ni.sparse_plan(mol, grids, True).nloc < -(-grids.size // 256)         # this rank holds only its tiles
    # analytic gradients with the aux-sharded tensor (every rank contracts its partial two-particle densities, (natm, 3)
    # all-reduce) against the same gradient from an unsharded tensor on this rank (shard override: no collectives)
    from pyscf_amd import scf
    for xc in ('', 'b3lyp'):
        mf = (dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)).density_fit(with_df=obj)
        if xc:
            mf.grids.level = 1
        mf.conv_tol = 1e-11
        mf.kernel()
        obj.grad_slab_bytes = 1 << 17       # r04: many Z slabs -> owners alternate between the ranks, every slab is reduced
        g_shard = mf.nuc_grad_method().kernel()
        assert obj._grad_slabs > 4, obj._grad_slabs
        full = df.DF(mol)
        full._shard_override = (0, 1)
        full.build()
        mf1 = (dft.RKS(mol, xc=xc) if xc else scf.RHF(mol)).density_fit(with_df=full)
        if xc:
            mf1.grids.level = 1
            mf1._numint._world_override = (0, 1)
        mf1.mo_coeff, mf1.mo_occ, mf1.mo_energy, mf1.e_tot, mf1.converged = mf.mo_coeff, mf.mo_occ, mf.mo_energy, mf.e_tot, True
        g_full = mf1.nuc_grad_method().kernel()
        assert np.abs(g_shard - g_full).max() < 1e-9, (xc, np.abs(g_shard - g_full).max())
        if not xc:                                   # (without grid response the XC part is not translationally invariant)
            assert np.abs(g_shard.sum(axis=0)).max() < 1e-8
    dist.barrier()
    if rank == 0:
        print('TWO_RANK_OK', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
