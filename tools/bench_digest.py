"""One-screen digest of a bench.py JSON line (GPU-box logs): python tools/bench_digest.py <file>"""
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print('no JSON line in', sys.argv[1], e)
    sys.exit(0)
r = d.get('roofline') or {}
keys = ('value', 'n_gpus', 'parts', 'value_host_api_ms')
print({k: d[k] for k in keys if k in d}, 'roofline', {k: r.get(k) for k in ('achieved', 'frac', 'ms', 'avg_launch_ms', 'traffic')},
      'traffic_source', str(r.get('traffic_source'))[:60])
print('  step', (d.get('roofline_step') or {}).get('frac'), 'cpu', {k: (d.get('cpu_baseline') or {}).get(k) for k in ('value', 'cores', 'kind')},
      'parity_golden', {k: (d.get('parity_golden') or {}).get(k) for k in ('ok', 'max_rel_err', 'golden')}, 'parity_sample', d.get('parity_sample'))
if d.get('comm'):
    c = d['comm']
    print('  comm', {k: c.get(k) for k in ('backend', 'comm_ms_per_step', 'bytes_per_step', 'peer_copies', 'bytes_per_part', 'push_ms_per_part',
                                         'sum_download_ms', 'compute_ms_per_part', 'algorithm_GBs_per_link') if k in c})
if d.get('kernels'):
    print('  kernels', {k: v['ms_total'] for k, v in d['kernels'].items()})
print('  layout', (d.get('config') or {}).get('tensor_layout'), 'hbm after build', (d.get('config') or {}).get('hbm_after_build_GB'), 'schedule', d.get('jk_schedule'), 'host calls', d.get('host_api_ms_calls'), (d.get('host_api_breakdown_ms') or [None])[0])
x = d.get('xc_path') or {}
print('  xc cached', (x.get('block_sparse') or {}).get('ao_cached_in_hbm'), (x.get('block_sparse') or {}).get('compact_ao_GB'))
print('  xc', x.get('nr_rks_ms_per_call'), x.get('kernels_ms'), {k: v.get('frac') for k, v in (x.get('roofline') or {}).items() if isinstance(v, dict) and 'frac' in v})
if (d.get('cpu_baseline') or {}).get('sample'):
    print('  cpu sample:', d['cpu_baseline']['sample'][-260:])
