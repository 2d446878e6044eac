#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_df_jk.py tests/test_gpu_fullsize_scf.py -x -q -m gpu --durations=5 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python tools/mfma_peak.py > $O/mfma_peak.log 2>&1; tail -1 $O/mfma_peak.log | cut -c1-700
timeout 600 python tools/prof_host_api.py > $O/prof_host_api.log 2>&1; head -4 $O/prof_host_api.log
timeout 900 python bench.py --molecule taxol --steps 5 --warmup 1 --no-cpu-baseline --xc '' > $O/bench_taxol.json 2> $O/bench_taxol.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03g/bench_taxol.json'))
print('taxol', d['value'], d['config']['workload'], {k:v['ms_total'] for k,v in d['kernels'].items()})
P
