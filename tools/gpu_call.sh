#!/bin/bash
# scratch: full GPU pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x --durations=8 -p no:warnings > gpurun_out/pytest_gpu_final.log 2>&1
tail -14 gpurun_out/pytest_gpu_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cut -c1-400 gpurun_out/bench_final.json
