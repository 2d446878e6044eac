#!/bin/bash
# round-3 GPU call A: new tests first, then the whole GPU suite, bench, rocprof stats
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_df_jk.py tests/test_gpu_bench_launch.py -x -q -m gpu > gpurun_out/r03a/pytest_new.log 2>&1
tail -5 gpurun_out/r03a/pytest_new.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
tail -c 3000 gpurun_out/r03a/bench.json; tail -5 gpurun_out/r03a/bench.err
timeout 1200 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r03a/pytest_gpu.log 2>&1
tail -20 gpurun_out/r03a/pytest_gpu.log
