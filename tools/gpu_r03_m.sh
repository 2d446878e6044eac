#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" --durations=3 > $O/pytest_c5.log 2>&1; tail -30 $O/pytest_c5.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_device_scf.py tests/test_gpu_native_abi.py -x -q -m gpu --durations=3 > $O/pytest_b.log 2>&1; tail -6 $O/pytest_b.log; cat gpurun_out/native_abi_worker.log | grep "s\]"
