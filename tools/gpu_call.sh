#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tdscf.py -m gpu -q -x -k exact > gpurun_out/pytest_td.log 2>&1
tail -15 gpurun_out/pytest_td.log
