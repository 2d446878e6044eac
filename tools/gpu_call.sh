cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_xc_sparse.py tests/test_gpu_response.py tests/test_gpu_tdscf.py tests/test_gpu_soscf.py tests/test_gpu_df_jk.py tests/test_gpu_scf.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/pytest_r02d.log
tail -6 gpurun_out/pytest_r02d.log
for w in 0 256 512 1024; do timeout 300 python tools/kbench.py --steps 3 --tag j2wg$w --tune j2wg=$w 2>/dev/null | tail -1; done > gpurun_out/kbench_r02d.log
cat gpurun_out/kbench_r02d.log
timeout 900 python tools/response_bench.py > gpurun_out/response_h2o32_r02.json 2> gpurun_out/response_h2o32_r02.err
tail -c 1200 gpurun_out/response_h2o32_r02.json; tail -3 gpurun_out/response_h2o32_r02.err
