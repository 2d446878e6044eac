#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dft.py tests/test_gpu_vhf.py tests/test_gpu_grad.py tests/test_gpu_response.py tests/test_gpu_xc_sparse.py -m gpu -q -x --durations=4 > gpurun_out/pytest_wb97.log 2>&1
tail -30 gpurun_out/pytest_wb97.log
