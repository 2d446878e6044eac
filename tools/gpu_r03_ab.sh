#!/bin/bash
# A/B on one box: fragment reads as single ds_read_b64 (product build) vs merged ds_read2_b64 (tools/ab build)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 6 "$@" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_frag_reads.log; }
for rep in 1 2; do
for v in b64 read2; do
  if [ $v = read2 ]; then export PAMD_LIBRARY=$PWD/tools/ab/libpyscf_amd_merged.so; else unset PAMD_LIBRARY; fi
  run --no-j --syrk-flags 12 --tag "$v K-only square, slots SYRK"
  run --no-j --syrk-flags 0 --tag "$v K-only square, plain SYRK"
  run --tag "$v J+K square"
  run --no-j --no-square --tag "$v K-only packed"
done
done
unset PAMD_LIBRARY
