#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# kernel trace + stats, then one PMC pass per counter set (never combined with --sys-trace etc.), outputs under
# gpurun_out/prof_<tag>*; copy the summaries into profiles/<tag>/ with tools/pmc_summarize.py afterwards.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --xc b3lyp"
mkdir -p $R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o $TAG -- $B > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o $TAG -- $B > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o $TAG -- $B > $R/gpurun_out/prof_${TAG}_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma -o $TAG -- $B > $R/gpurun_out/prof_${TAG}_mfma.log 2>&1
cd $R
tail -1 gpurun_out/prof_${TAG}_bench.log | cut -c1-400
find gpurun_out/prof_${TAG}* -name "*.csv" | head -20
