#!/bin/bash
# packed-operand half transform with the diagonal-block side image: parity tests, then A/B at config-3 shapes
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -q -m gpu -x -k "packed or partial_square or mo_branch" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for t in pkdiag=0 pkdiag=1; do
  timeout 300 python tools/kbench.py --steps 5 --no-square --tune $t --tag "J+K packed $t" 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/kbench_pkdiag.log
  timeout 300 python tools/kbench.py --steps 5 --no-square --no-j --tune $t --tag "K-only packed $t" 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/kbench_pkdiag.log
done
timeout 300 python tools/kbench.py --nao 2228 --naux 1400 --nocc 226 --steps 3 --no-square --tune pkdiag=0 --tag "taxol-quarter packed pkdiag=0" 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/kbench_pkdiag.log
timeout 300 python tools/kbench.py --nao 2228 --naux 1400 --nocc 226 --steps 3 --no-square --tune pkdiag=1 --tag "taxol-quarter packed pkdiag=1" 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/kbench_pkdiag.log
