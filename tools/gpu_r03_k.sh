#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03k; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest_gpu.log 2>&1; tail -18 $O/pytest_gpu.log
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log | cut -c1-300
timeout 300 python tools/run_scf.py --nwater 32 --xc b3lyp --conv-tol 1e-10 > $O/scf_h2o32_b3lyp.log 2>&1; tail -4 $O/scf_h2o32_b3lyp.log
