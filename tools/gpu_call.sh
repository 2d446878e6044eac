#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for t in "vmatprobe=0" "vmatprobe=1"; do
timeout 400 python tools/xcbench.py --steps 3 --tune-xc $t 2>/dev/null | tail -1 | cut -c1-500
done
