#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for t in "gemmwide=1" "gemmwide=0"; do
timeout 400 python tools/response_bench.py --skip-general --tune $t 2>/dev/null | tail -1 | cut -c1-420
done
