#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03handle; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_native_abi.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/native_bench.py > $O/native_bench.log 2>&1; tail -5 $O/native_bench.log
