#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in "syrksk=1" "syrksk=0" "syrksk=1" "syrksk=0"; do
timeout 300 python tools/kbench.py --steps 4 --tune $t --tag "$t-Konly" --no-j 2>/dev/null | tail -1 | cut -c1-330
done
