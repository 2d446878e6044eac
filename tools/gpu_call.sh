cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/pytest_r02n.log
tail -4 gpurun_out/pytest_r02n.log
for t in "pairtail=0" "pairtail=1"; do timeout 300 python tools/kbench.py --steps 3 --tag $t --tune $t 2>/dev/null | tail -1 | cut -c1-330; done > gpurun_out/kbench_r02n.log
cat gpurun_out/kbench_r02n.log
