"""world_size-2 gloo test (CPU): aux-index sharding + all-reduce of the partial J/K reproduces
the single-process result.  The per-shard arithmetic is done by the oracle here (no GPU in this
tier); what is under test is the host logic of the N>1 path: DF.shard_range and
df_jk._allreduce over a torch.distributed group."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import H2O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import ref
    from pyscf_amd import gto, df
    from pyscf_amd.df import df_jk
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    aux = df.make_auxmol(mol, 'weigend')
    cderi = ref.cholesky_eri(mol, aux)
    obj = df.DF(mol, 'weigend')
    assert obj.world_size == world and obj.rank == rank
    l0, l1 = obj.shard_range(cderi.shape[0], rank, world)
    np.random.seed(1)
    dms = np.random.random((2, mol.nao, mol.nao))
    vj, vk = ref.get_jk(cderi[l0:l1], dms, hermi=0)           # partial J/K of this shard
    tj, tk = torch.from_numpy(vj.copy()), torch.from_numpy(vk.copy())
    df_jk._allreduce(obj, [tj, tk])
    if rank == 0:
        q.put((ref.fp(tj.numpy()), ref.fp(tk.numpy())))
    dist.destroy_process_group()


def test_two_rank_aux_sharding_allreduce():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    fj, fk = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # golden fingerprints of the full (unsharded) J/K: pyscf/df/test/test_df_jk.py:144-152
    assert abs(fj - -194.15910890730066) < 1e-9
    assert abs(fk - -46.365071587653517) < 1e-9


def _worker_host_partials(rank, world, port, q):
    """r05: the host-array all-reduce of an out-of-core rank (DF._allreduce_host): a stand-in for the C handle returns this rank's
    partial J/K as numpy arrays, DF.get_jk must sum them over the gloo group in place."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import ref
    from pyscf_amd import gto, df
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    aux = df.make_auxmol(mol, 'weigend')
    cderi = ref.cholesky_eri(mol, aux)
    obj = df.DF(mol, 'weigend')
    l0, l1 = obj.shard_range(cderi.shape[0], rank, world)

    class _Handle:                       # what NativeDF(shard=(rank, world)) answers: partial sums of its rows, host arrays
        shard = (rank, world)
        shard_rows = (l0, l1)

        def get_jk(self, dm, hermi, with_j, with_k, tol):
            vj, vk = ref.get_jk(cderi[l0:l1], np.asarray(dm), hermi=hermi, with_j=with_j, with_k=with_k)
            return vj, vk
    obj._native = _Handle()
    obj._naux = cderi.shape[0]
    np.random.seed(1)
    dms = np.random.random((2, mol.nao, mol.nao))
    vj, vk = obj.get_jk(dms, hermi=0)
    vj_only, none = obj.get_jk(dms, hermi=0, with_k=False)
    assert none is None and np.abs(vj_only - vj).max() < 1e-12
    if rank == 0:
        q.put((ref.fp(vj), ref.fp(vk)))
    dist.destroy_process_group()


def test_two_rank_out_of_core_partials_are_summed_on_the_host():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_host_partials, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    fj, fk = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(fj - -194.15910890730066) < 1e-9
    assert abs(fk - -46.365071587653517) < 1e-9
