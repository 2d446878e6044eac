cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py tests/test_gpu_xc_sparse.py tests/test_gpu_bench_launch.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_r02b.log
tail -5 gpurun_out/pytest_r02b.log
for t in "dmav2=0" "dmav2=1"; do timeout 300 python tools/kbench.py --steps 3 --tag $t --tune $t 2>/dev/null | tail -1; done > gpurun_out/kbench_r02b.log
cat gpurun_out/kbench_r02b.log
timeout 600 python bench.py --steps 5 --warmup 1 --cpu-threads 64 > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02b.json'))
print(d['value'], d['value_host_api_ms'], d['roofline']['achieved'], d['kernels'], d['cpu_baseline'])
PY
