#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in "" "--j2-defer" "" "--j2-defer"; do
timeout 300 python tools/kbench.py --steps 5 --tag "defer:$t" $t 2>/dev/null | tail -1 | cut -c1-330
done
