#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03suite; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=6 > $O/pytest_gpu.log 2>&1; tail -10 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
