#!/bin/bash
# last state of the round: whole GPU suite in one go, default bench line, the same without the square image, smoke
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest_gpu.log 2>&1; tail -18 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
timeout 600 python bench.py --k-square off --no-cpu-baseline --xc '' > $O/bench_ksquare_off.json 2> $O/bench_ksquare_off.err; cut -c1-200 $O/bench_ksquare_off.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
