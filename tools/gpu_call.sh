cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -6 | cut -c1-250
for t in "syrktall=0" "syrktall=1"; do timeout 300 python tools/kbench.py --steps 3 --tag $t --tune $t 2>/dev/null | tail -1 | cut -c1-330; done > gpurun_out/kbench_r02s.log
timeout 300 python tools/kbench.py --steps 3 --tag ksplit4_square --ksplit 4 --tune syrktall=0 2>/dev/null | tail -1 | cut -c1-330 >> gpurun_out/kbench_r02s.log
cat gpurun_out/kbench_r02s.log
