#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03i; mkdir -p $O
for t in "syrkprobe=0" "syrkprobe=1"; do
  timeout 300 python tools/kbench.py --steps 5 --syrk-flags 12 --tune $t --tag "J+K syrk-flags=12 $t" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_syrk_dma_probe.log
  timeout 300 python tools/kbench.py --steps 5 --no-j --syrk-flags 12 --tune $t --tag "K-only syrk-flags=12 $t" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_syrk_dma_probe.log
done
timeout 300 python tools/kbench.py --steps 5 --syrk-flags 0 --tag "J+K syrk-flags=0" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_syrk_dma_probe.log
