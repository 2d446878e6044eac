cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/pytest_r02o.log
tail -2 gpurun_out/pytest_r02o.log
for t in "dmafront=0,xcdmap=0" "dmafront=1,xcdmap=0" "dmafront=0,xcdmap=1" "dmafront=1,xcdmap=1"; do timeout 300 python tools/kbench.py --steps 3 --no-overlap --tag $t --tune $t 2>/dev/null | tail -1 | cut -c1-330; done > gpurun_out/kbench_r02o.log
cat gpurun_out/kbench_r02o.log
for t in "dmafront=0" "dmafront=1"; do timeout 300 python tools/kbench.py --steps 3 --no-overlap --no-square --tag nosq_$t --tune $t 2>/dev/null | tail -1 | cut -c1-330; done >> gpurun_out/kbench_r02o.log
tail -2 gpurun_out/kbench_r02o.log
