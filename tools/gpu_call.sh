#!/bin/bash
# scratch: full GPU pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/pytest_gpu_final.log 2>&1
tail -20 gpurun_out/pytest_gpu_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json | cut -c1-900
bash tools/profile_round.sh r02 > /dev/null 2>&1
ls gpurun_out/prof_r02/ | head
