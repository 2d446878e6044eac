#!/bin/bash
# e2-first sub-block pipeline: J pass 2 of sub-block s beside the half transform of s+1; one SYRK per block
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_df_jk.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 6 "$@" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_e2_pipeline.log; }
run --tag "J+K default (2 blocks of 8 GB, SYRK+J2 each)"
run --block-gb 12 --tag "J+K one block, pipeline 1"
for n in 2 3 4 6; do
run --block-gb 12 --e2-pipeline $n --tag "J+K one block, e2 pipeline $n, syrk plain"
run --block-gb 12 --e2-pipeline $n --syrk-flags 12 --tag "J+K one block, e2 pipeline $n, syrk flags 12"
done
run --e2-pipeline 2 --syrk-flags 12 --tag "J+K 2 blocks, e2 pipeline 2, syrk flags 12"
run --tag "J+K default (again)"
