#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03t2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_df_jk.py tests/test_gpu_native_abi.py tests/test_gpu_cabi_kernels.py tests/test_gpu_response.py tests/test_gpu_fullsize.py -q -m gpu -x --durations=8 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
