"""RCCL on the GPU box: nobody had ever executed `init_process_group('nccl', device_id=...)`, the f64 packed
all-reduce or `dist.barrier()` on RCCL.  A 1-GPU box can still run all of them with world_size = 1 and the collective path
forced (pyscf_amd.lib.comm.force).  Serial decomposition replaced: pyscf/df/df_jk.py:362-381."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return str(sk.getsockname()[1])


@pytest.mark.gpu
def test_rccl_world_size_one_collectives_change_nothing():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['MASTER_ADDR'] = '127.0.0.1'
    env['MASTER_PORT'] = _free_port()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_world1.py')], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0 and 'RCCL_WORLD1_OK' in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.gpu
def test_bench_one_rank_under_torchrun_rccl():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, RCCL backend), with N = 1 and the
    collectives forced: the JSON line must come out with comm_ms measured and the same J/K checksum as a plain run."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['PAMD_FORCE_COLLECTIVE'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', _free_port(), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--nwater', '4', '--basis', 'cc-pvdz', '--no-cpu-baseline', '--xc', '']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    import json
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 1 and out['comm']['backend'] == 'nccl' and out['comm']['collectives_per_step'] >= 1
    assert out['comm']['comm_ms_per_step'] is not None and out['comm']['comm_ms_per_step'] >= 0
