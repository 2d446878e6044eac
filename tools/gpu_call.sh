#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "config4" --durations=3 > gpurun_out/pytest_c4.log 2>&1
tail -8 gpurun_out/pytest_c4.log
