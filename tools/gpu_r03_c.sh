#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_device_scf.py tests/test_gpu_native_abi.py -x -q -m gpu --durations=8 > $O/pytest_new.log 2>&1
tail -25 $O/pytest_new.log
timeout 900 python tools/run_scf.py --molecule taxol --xc '' --conv-tol 1e-10 --max-cycle 60 --dump-orbitals gpurun_out/taxol_rhf_orbitals.npz > $O/scf_taxol_rhf.log 2>&1; tail -4 $O/scf_taxol_rhf.log
timeout 600 python tools/prof_host_api.py > $O/prof_host_api.log 2>&1; head -40 $O/prof_host_api.log
timeout 600 python -m cProfile -s cumulative tools/run_scf.py --nwater 32 --xc b3lyp --conv-tol 1e-10 > $O/prof_scf_b3lyp.log 2>&1; grep -v "^cycle\|nelec" $O/prof_scf_b3lyp.log | head -70
