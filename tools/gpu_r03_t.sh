#!/bin/bash
# packed path at step level: where do the 9.7 ms between K-only and J+K go?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03t; mkdir -p $O
run() { timeout 300 python tools/kbench.py --steps 5 --no-square "$@" 2>/dev/null | tail -1 | cut -c1-520 | tee -a $O/kbench_packed_step.log; }
run --tag "J+K packed (default)"
run --no-fuse --tag "J+K packed, two-pass J (no fused pass 1)"
run --syrk-flags 12 --tag "J+K packed, syrk-flags=12"
run --no-overlap --tag "J+K packed, J not overlapped"
run --no-overlap --syrk-flags 12 --tag "J+K packed, J not overlapped, syrk-flags=12"
timeout 600 python bench.py --k-square off --no-cpu-baseline --xc '' > $O/bench_ksquare_off.json 2> $O/bench_ksquare_off.err; cut -c1-900 $O/bench_ksquare_off.json; tail -2 $O/bench_ksquare_off.err
