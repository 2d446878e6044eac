"""bench.py starts its own ranks: `python bench.py --gpus 2` must produce an n_gpus = 2 line whose ranks hold disjoint
aux-row shards (two ranks share the one device of the test box through gloo; on a multi-GPU node the same launch uses
RCCL, one device per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--nwater', '8', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', '--xc', ''] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    return json.loads(line)


def test_bench_gpus2_launches_two_ranks():
    one = _run(['--gpus', '1'])
    two = _run(['--gpus', '2', '--backend', 'gloo'])
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    rows = two['config']['naux_per_rank']
    assert len(rows) == 2 and sum(rows) == one['config']['naux_per_rank'][0] and abs(rows[0] - rows[1]) <= 1
    assert two['config']['naux_local'] == rows[0]
    assert two['value'] > 0 and two['value_host_api_ms'] > 0
    # the driver's contract for the JSON line
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in one, key
    assert one['steps'] == 2 and one['warmup'] == 1 and one['higher_is_better'] is False and one['dtype'] == 'f64'
    assert one['unit'] == 'ms' and abs(one['value'] - one['ms_per_step']) < 1e-9 and 'workload' in one['config']
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(one['roofline'])
    assert abs(one['roofline']['frac'] - one['roofline']['achieved'] / one['roofline']['peak']) < 1e-3
    assert one['jk_schedule']['chosen'] in ('overlap', 'serial', 'fused', 'auto')
    # the roofline durations are those of the timed steps (live HIP events), the serial pass is reported beside them
    assert one['roofline']['launches_per_step'] >= 1 and one['kernels_serial_pass'] and 'e2_symm' in one['kernels']
    assert one['kernels']['e2_symm']['ms_total'] < one['ms_per_step']
    # r05: in-run correctness evidence and live HBM counters are part of every line (the golden exists for configs 3 / 4 only)
    for d in (one, two):
        assert 'parity_golden' in d and d['parity_golden']['golden'] is None and 'why' in d['parity_golden']
    import shutil
    if shutil.which('rocprofv3') and one['roofline']['traffic'] is not None:
        assert 'measured in this run' in one['roofline']['traffic_source'], one['roofline']['traffic_source']


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--nwater', '2'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and 'WORLD_SIZE' in (out.stderr + out.stdout)


def test_two_rank_product_path():
    """The N > 1 arithmetic through the PRODUCT on the GPU (two gloo ranks on one device): sharded tensor build, J/K of
    both branches, the rank-local save / reload, round-robin grid tiles of nr_rks - all against the CPU oracle."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(ROOT, 'tests', '_two_rank_product.py')], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and 'TWO_RANK_OK' in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
