#!/bin/bash
# after the oracle's independent B3LYP golden: the config-3 SCF tests, taxol on one GPU, then the profile passes of the final bench
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03last; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize_scf.py -q -m gpu -x > $O/pytest_scf.log 2>&1; tail -3 $O/pytest_scf.log
timeout 900 python bench.py --molecule taxol --no-cpu-baseline --xc '' --steps 3 > $O/bench_taxol_1gpu.json 2> $O/bench_taxol_1gpu.err; cut -c1-140 $O/bench_taxol_1gpu.json
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
