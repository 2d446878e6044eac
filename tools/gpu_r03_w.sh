#!/bin/bash
# e2_pk with L2 prefetch of tile v + pfd: parity, then A/B of the distance on one box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -q -m gpu -x -k "packed or partial_square or mo_branch" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 6 --no-square "$@" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_pk_prefetch.log; }
for t in pkpf=0 pkpf=1 pkpf=2 pkpf=3 pkpf=4 pkpf=0 pkpf=2; do
run --no-j --tune $t --tag "K-only packed $t"
done
for t in pkpf=0 pkpf=2 pkpf=3; do
run --tune $t --tag "J+K packed $t"
done
timeout 300 python tools/kbench.py --steps 6 --tag "J+K square" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_pk_prefetch.log
