#!/bin/bash
# e2_pk DIAG: compile-time layout of the staged tile (product build) vs run-time selects (previous build in tools/ab)
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r03ab3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -q -m gpu -x -k "packed or partial_square or mo_branch or ragged" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 6 --no-square "$@" 2>/dev/null | tail -1 | cut -c1-250 | tee -a $O/kbench_pk_ntr.log; }
for rep in 1 2; do
  export PAMD_LIBRARY=$PWD/tools/ab/libpyscf_amd_ab.so; run --no-j --tag "prev K-only packed"; run --tag "prev J+K packed"
  unset PAMD_LIBRARY; run --no-j --tag "new K-only packed"; run --tag "new J+K packed"
done
run --nao 2228 --naux 1400 --nocc 226 --steps 3 --no-j --tag "new taxol-quarter K-only packed"
export PAMD_LIBRARY=$PWD/tools/ab/libpyscf_amd_ab.so; run --nao 2228 --naux 1400 --nocc 226 --steps 3 --no-j --tag "prev taxol-quarter K-only packed"
