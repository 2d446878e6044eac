#!/bin/bash
# A/B on one box: rho epilogue reading the transposed orbital copy (rhoorbt=1) or the orbitals as they are (0)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03v; mkdir -p $O
run() { timeout 300 python tools/kbench.py --steps 8 "$@" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_rho_epilogue.log; }
for rep in 1 2; do
for t in rhoorbt=0 rhoorbt=1; do
run --tune $t --tag "J+K square $t"
run --no-square --tune $t --tag "J+K packed $t"
done
done
