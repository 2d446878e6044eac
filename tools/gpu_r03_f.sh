#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_df_jk.py -x -q -m gpu > $O/pytest_jk.log 2>&1; tail -3 $O/pytest_jk.log
for f in 0 12; do
  timeout 300 python tools/kbench.py --steps 5 --syrk-flags $f --tag "J+K syrk-flags=$f" 2>/dev/null | tail -1 | cut -c1-400 | tee -a $O/kbench_syrk_jk.log
done
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03f/bench.json'))
print(d['value'], d['value_host_api_ms'], json.dumps(d['roofline_step'])[:300], d['kernels']['dgemm_tn'], d['kernels']['e2_symm'])
P
timeout 900 python bench.py --molecule taxol --steps 5 --warmup 1 --no-cpu-baseline --xc '' > $O/bench_taxol.json 2> $O/bench_taxol.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03f/bench_taxol.json'))
print('taxol', d['value'], d['kernels'])
P
