cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/uks_scale.py > gpurun_out/uks_h2o32_cation.log 2>&1
tail -1 gpurun_out/uks_h2o32_cation.log | cut -c1-700
timeout 600 python tools/grad_bench.py --nwater 32 > gpurun_out/grad_h2o32_rhf_r02.json 2> gpurun_out/grad_r02.err
tail -1 gpurun_out/grad_h2o32_rhf_r02.json; tail -2 gpurun_out/grad_r02.err
timeout 600 python tools/grad_bench.py --nwater 32 --xc b3lyp --grid-response > gpurun_out/grad_h2o32_b3lyp_r02.json 2>> gpurun_out/grad_r02.err
tail -1 gpurun_out/grad_h2o32_b3lyp_r02.json
