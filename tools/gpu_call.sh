#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vhf.py -m gpu -q -x --durations=8 > gpurun_out/pytest_vhf.log 2>&1
tail -40 gpurun_out/pytest_vhf.log
