#!/bin/bash
# final run of the round: whole GPU suite (with the config-4 / config-5 goldens), default bench line, smoke
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03q; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest_gpu.log 2>&1; tail -18 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
