#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_df_jk.py tests/test_gpu_vhf.py -x -q -m gpu > $O/pytest_jk_vhf.log 2>&1; tail -12 $O/pytest_jk_vhf.log
for f in 0 4 8 12; do
  timeout 300 python tools/kbench.py --steps 5 --no-j --syrk-flags $f --tag "K-only syrk-flags=$f" 2>/dev/null | tail -1 | cut -c1-400 | tee -a $O/kbench_syrk_variants.log
done
for f in 0 8; do
  timeout 300 python tools/kbench.py --steps 5 --syrk-flags $f --tag "J+K syrk-flags=$f" 2>/dev/null | tail -1 | cut -c1-400 | tee -a $O/kbench_syrk_variants.log
done
timeout 300 python tools/mfma_peak.py > $O/mfma_peak.log 2>&1; tail -2 $O/mfma_peak.log | cut -c1-900
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03e/bench.json'))
print(d['value'], d['value_host_api_ms'], json.dumps(d['roofline_step']), d['kernels']['dgemm_tn'], d['kernels']['e2_symm'])
P
