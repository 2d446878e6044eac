#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_int3c2e.py tests/test_gpu_scf.py -m gpu -q -x --durations=5 > gpurun_out/pytest_tm.log 2>&1
tail -25 gpurun_out/pytest_tm.log
