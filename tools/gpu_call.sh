cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_xc_sparse.py tests/test_gpu_dft.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/pytest_r02k.log
tail -5 gpurun_out/pytest_r02k.log
timeout 300 python tools/xcbench.py 2>/dev/null | tail -1 > gpurun_out/xcbench_r02k.json
cut -c1-600 gpurun_out/xcbench_r02k.json
