#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vhf.py tests/test_gpu_df_jk.py -m gpu -q -x --durations=5 > gpurun_out/pytest_gold.log 2>&1
tail -15 gpurun_out/pytest_gold.log
