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
        assert d['parity_golden']['golden'] == 'tests/golden/h2o8_ccpvtz_oracle.json' and d['parity_golden']['ok'] is True
        assert d['parity_golden']['max_rel_err'] < 1e-9
    import shutil
    if shutil.which('rocprofv3') and one['roofline']['traffic'] is not None:
        assert 'measured in this run' in one['roofline']['traffic_source'], one['roofline']['traffic_source']


def test_bench_two_rank_line_is_complete():
    """r06: an N > 1 line carries everything an N = 1 line does - `roofline` (with the traffic of rank 0's
    shard MEASURED in the run under --pmc on, or scaled and labelled so), `cpu_baseline` (the reference C on rank 0's shard rows,
    shard value + the extrapolation to all rows, labelled), `comm`, `parity_golden` - before it is ever run on 8 GPUs."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    import shutil
    have_prof = shutil.which('rocprofv3') is not None
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--nwater', '8', '--steps', '2', '--warmup', '1', '--xc', '',
                          '--gpus', '2', '--backend', 'gloo', '--cpu-threads', '16'] + (['--pmc', 'on'] if have_prof else []),
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    two = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert two['n_gpus'] == 2 and len(two['config']['naux_per_rank']) == 2
    r = two['roofline']
    assert r is not None and r['frac'] > 0 and 'traffic_measured_in_run' in r and r['traffic_source']
    if have_prof and r['traffic'] is not None and r['traffic_measured_in_run']:
        assert 'measured in this run' in r['traffic_source']
    elif r['traffic'] is not None:
        assert 'SCALED' in r['traffic_source']
    c = two['cpu_baseline']
    assert c is not None and c['value'] > 0 and c['kind'] in ('reference', 'port') and c['cores'] >= 1
    assert c['shard_rows'] == two['config']['naux_per_rank'][0] and c['shard_value'] > 0 and 'extrapolation' in c
    assert abs(c['value'] - c['shard_value'] * sum(two['config']['naux_per_rank']) / c['shard_rows']) < 0.2 + 1e-3 * c['value']
    assert two['comm'] is not None and two['comm']['backend'] == 'gloo' and two['comm']['collectives_per_step'] >= 1
    assert two['comm']['bytes_per_step'] > 0
    assert two['parity_golden']['ok'] is True and two['parity_sample'] is not None
    assert two['parity_sample']['max_abs_err_vk'] < 1e-8 and two['parity_sample']['max_abs_err_vj'] < 1e-8
    assert isinstance(two['host_api_breakdown_ms'], list) and two['host_api_breakdown_ms'] and 'probe' in two['host_api_breakdown_ms'][0]


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
