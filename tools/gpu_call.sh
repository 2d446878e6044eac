#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in "syrk3=0 --ksplit 4" "syrk3=1 --ksplit 4" "syrk3=1 --ksplit 6" "syrk3=1 --ksplit 3"; do
timeout 300 python tools/kbench.py --steps 4 --no-j --tag "$t" --tune $t 2>/dev/null | tail -1 | cut -c1-250
done
for t in "syrk3=0 --ksplit 4" "syrk3=1 --ksplit 6"; do
timeout 300 python tools/kbench.py --steps 4 --tag "JK $t" --tune $t 2>/dev/null | tail -1 | cut -c1-300
done
