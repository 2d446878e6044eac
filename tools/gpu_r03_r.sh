#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03r; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fullsize_cfg45.py tests/test_gpu_fullsize_scf.py tests/test_gpu_grad.py tests/test_gpu_int3c2e.py tests/test_gpu_native_abi.py tests/test_gpu_rccl.py tests/test_gpu_response.py tests/test_gpu_scf.py tests/test_gpu_soscf.py tests/test_gpu_tdscf.py tests/test_gpu_vhf.py tests/test_gpu_xc_sparse.py -q -m gpu -x --durations=10 > $O/pytest_rest.log 2>&1; tail -16 $O/pytest_rest.log
