cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize_scf.py 2>&1 | tail -40 > gpurun_out/pytest_r02e.log
tail -8 gpurun_out/pytest_r02e.log
timeout 300 python tools/run_scf.py --nwater 32 --xc b3lyp > gpurun_out/scf_h2o32_b3lyp_r02e.log 2>&1
grep "cycle= 6\|cycle= 7\|converged" gpurun_out/scf_h2o32_b3lyp_r02e.log
for t in 512 2048; do timeout 300 python tools/xcbench.py --tile $t 2>/dev/null | tail -1; done > gpurun_out/xcbench_tiles.log
cut -c1-700 gpurun_out/xcbench_tiles.log
