cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_r02p.log
tail -6 gpurun_out/pytest_r02p.log
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r02p.json 2> gpurun_out/bench_r02p.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02p.json'))
print(d['value'], d['value_host_api_ms'], d['roofline']['achieved'], d['roofline']['frac'], {k:v['ms_total'] for k,v in d['kernels'].items()}, d['cpu_baseline']['value'], d['xc_path']['nr_rks_ms_per_call'])
PY
