#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03o; mkdir -p $O
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --xc '' "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['kernels']['dgemm_tn']['ms_total'], d['kernels']['e2_symm']['ms_total'])" | tee -a $O/bench_ab.log; }
run --syrk-flags 0
run --syrk-flags 12 --tune mfmaprio=1
run --syrk-flags 12 --tune mfmaprio=1,e2prio=1
run --syrk-flags 0 --tune mfmaprio=1,e2prio=1
run --syrk-flags 12
run --syrk-flags 0
