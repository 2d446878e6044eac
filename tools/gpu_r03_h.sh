#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03h; mkdir -p $O
for t in "j2wg=0" "j2wg=64" "j2wg=256" "j2wg=1024"; do
  timeout 300 python tools/kbench.py --steps 5 --syrk-flags 12 --tune $t --tag "J+K syrk-flags=12 $t" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_syrk12_j2wg.log
done
for t in "j2wg=64" "j2wg=256"; do
  timeout 300 python tools/kbench.py --steps 5 --syrk-flags 0 --tune $t --tag "J+K syrk-flags=0 $t" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_syrk12_j2wg.log
done
