#!/bin/bash
# fused first J pass with the transposed-orbital (coalesced) epilogue: parity, then step level, square and packed
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py tests/test_gpu_native_abi.py -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 5 "$@" 2>/dev/null | tail -1 | cut -c1-420 | tee -a $O/kbench_rho_epilogue.log; }
run --tag "J+K square"
run --no-square --tag "J+K packed"
run --tag "J+K square (again)"
run --no-square --tag "J+K packed (again)"
timeout 600 python bench.py --no-cpu-baseline --xc '' > $O/bench_square.json 2> $O/bench_square.err; cut -c1-420 $O/bench_square.json; tail -2 $O/bench_square.err
timeout 600 python bench.py --k-square off --no-cpu-baseline --xc '' > $O/bench_ksquare_off.json 2> $O/bench_ksquare_off.err; cut -c1-420 $O/bench_ksquare_off.json; tail -2 $O/bench_ksquare_off.err
