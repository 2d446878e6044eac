#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python tests/_native_abi_worker.py > $O/native_abi.log 2>&1; tail -6 $O/native_abi.log
timeout 600 python -m pytest tests/test_gpu_native_abi.py tests/test_gpu_vhf.py tests/test_gpu_rccl.py -x -q -m gpu --durations=5 > $O/pytest.log 2>&1; tail -9 $O/pytest.log
