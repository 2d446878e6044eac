cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ls /opt/conda/lib/libhdf5.so* | head -2
timeout 900 python -m pytest tests/test_gpu_bench_launch.py tests/test_gpu_df_jk.py tests/test_gpu_scf.py -m gpu -q -x 2>&1 | tail -12 | cut -c1-250
