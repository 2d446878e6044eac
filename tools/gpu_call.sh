#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in "syrkprobe=0" "syrkprobe=1" "syrkprobe=2"; do
timeout 300 python tools/kbench.py --steps 4 --tune $t --tag "$t" --no-j 2>/dev/null | tail -1 | cut -c1-260
done
