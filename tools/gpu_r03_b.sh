#!/bin/bash
# round-3 GPU call B: device-resident SCF loop, reference CPU baseline, taxol orbitals for the oracle
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_device_scf.py tests/test_gpu_fullsize_scf.py tests/test_gpu_scf.py -x -q -m gpu --durations=8 > $O/pytest_scf.log 2>&1
tail -15 $O/pytest_scf.log
timeout 600 python tools/run_scf.py --nwater 32 --xc b3lyp --conv-tol 1e-10 > $O/scf_h2o32_b3lyp.log 2>&1; tail -22 $O/scf_h2o32_b3lyp.log
timeout 600 python tools/run_scf.py --nwater 32 --xc '' --conv-tol 1e-10 > $O/scf_h2o32_rhf.log 2>&1; tail -5 $O/scf_h2o32_rhf.log
timeout 900 python tools/run_scf.py --molecule taxol --xc '' --conv-tol 1e-10 --max-cycle 60 --dump-orbitals gpurun_out/taxol_rhf_orbitals.npz > $O/scf_taxol_rhf.log 2>&1; tail -8 $O/scf_taxol_rhf.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -c 2500 $O/bench.json; tail -5 $O/bench.err
