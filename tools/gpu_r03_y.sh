#!/bin/bash
# native handle with the diagonal-block image, bench with live kernel events, taxol on one GPU
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_native_abi.py tests/test_gpu_bench_launch.py -q -m gpu -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 900 python bench.py --molecule taxol --no-cpu-baseline --xc '' --steps 3 > $O/bench_taxol_1gpu.json 2> $O/bench_taxol_1gpu.err; cut -c1-700 $O/bench_taxol_1gpu.json; tail -2 $O/bench_taxol_1gpu.err
python - <<'P'
import json
d = json.load(open('gpurun_out/r03y/bench_taxol_1gpu.json'))
print(d['value'], d['kernels'], d.get('kernels_serial_pass'))
P
