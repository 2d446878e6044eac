#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bench_launch.py -m gpu -q -x > gpurun_out/pytest_2rank.log 2>&1
tail -6 gpurun_out/pytest_2rank.log
