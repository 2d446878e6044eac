cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cabi_kernels.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/pytest_r02j.log
tail -4 gpurun_out/pytest_r02j.log
for t in "pkdma=0" "pkdma=1"; do timeout 300 python tools/kbench.py --steps 3 --no-square --no-overlap --tag $t --tune $t 2>/dev/null | tail -1 | cut -c1-260; done > gpurun_out/kbench_r02j.log
cat gpurun_out/kbench_r02j.log
