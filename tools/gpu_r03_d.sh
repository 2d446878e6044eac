#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03d; mkdir -p $O
for t in "syrkfrac=0" "syrkfrac=1"; do
  timeout 300 python tools/kbench.py --steps 5 --tune $t --tag $t 2>/dev/null | tail -1 | cut -c1-600 | tee -a $O/kbench_syrk_balanced.log
done
for t in "syrkfrac=0" "syrkfrac=1"; do
  timeout 300 python tools/kbench.py --steps 5 --no-j --tune $t --tag "K-only $t" 2>/dev/null | tail -1 | cut -c1-600 | tee -a $O/kbench_syrk_balanced.log
done
for t in "syrkfrac=0" "syrkfrac=1"; do
  timeout 300 python tools/kbench.py --nao 2228 --naux 1400 --nocc 226 --steps 3 --tune $t --tag "taxol-quarter $t" 2>/dev/null | tail -1 | cut -c1-600 | tee -a $O/kbench_syrk_balanced.log
done
timeout 600 python -m pytest tests/test_gpu_df_jk.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_jk.log 2>&1; tail -4 $O/pytest_jk.log
timeout 600 python tests/_native_abi_worker.py > $O/native_abi.log 2>&1; tail -6 $O/native_abi.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03d/bench.json'))
print(d['value'], d['value_host_api_ms'], d['roofline_step'], d['kernels'])
P
