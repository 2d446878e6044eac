#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --xc '' > gpurun_out/bench_h2o32_na$i.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('gpurun_out/bench_h2o32_na$i.json').read().strip().splitlines()[-1])
print(d['value'], d['value_host_api_ms'], {k:(round(v['ms_total'],1),v['launches']) for k,v in d['kernels'].items()})
PY
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
