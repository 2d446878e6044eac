#!/bin/bash
# scratch: QZ / second-row tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_int3c2e.py tests/test_gpu_scf.py "tests/test_gpu_grad.py::test_df_rhf_gradient_higher_l_vs_fd" -m gpu -q -x --durations=8 > gpurun_out/pytest_qz.log 2>&1
tail -25 gpurun_out/pytest_qz.log
