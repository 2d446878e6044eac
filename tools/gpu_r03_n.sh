#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03n; mkdir -p $O
for f in 0 12; do
for t in "j2wide=0" "j2wide=1" "mfmaprio=1" "j2wide=1,mfmaprio=1"; do
  timeout 300 python tools/kbench.py --steps 5 --syrk-flags $f --tune $t --tag "J+K syrk-flags=$f $t" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_j2wide_prio.log
done
done
timeout 300 python -m pytest tests/test_gpu_native_abi.py tests/test_gpu_df_jk.py -x -q -m gpu --durations=3 2>&1 | tail -6
