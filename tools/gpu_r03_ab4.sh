#!/bin/bash
# e2_sq2: pair-tail workgroups inside the main launch (e2merge=1) vs a second launch (0)
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r03ab4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 8 "$@" 2>/dev/null | tail -1 | cut -c1-250 | tee -a $O/kbench_e2merge.log; }
for rep in 1 2; do
for t in e2merge=0 e2merge=1; do
  run --no-j --syrk-flags 12 --tune $t --tag "K-only $t"
  run --tune $t --tag "J+K $t"
done
done
