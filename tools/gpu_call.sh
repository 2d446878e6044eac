cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_r02m.log
tail -8 gpurun_out/pytest_r02m.log
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r02m.json 2> gpurun_out/bench_r02m.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02m.json'))
print(d['value'], d['value_host_api_ms'], d['roofline']['achieved'], {k:v['ms_total'] for k,v in d['kernels'].items()}, d['cpu_baseline']['value'], d['cpu_baseline']['phases_s'], d['xc_path']['nr_rks_ms_per_call'], d['xc_path']['kernels_ms'])
PY
timeout 900 bash tools/profile_round.sh r02 2>&1 | tail -3
