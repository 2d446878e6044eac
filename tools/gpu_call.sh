#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/shard_h2o32.jsonl
for w in 1 2 4 8; do
timeout 300 python tools/shard_probe.py --nwater 32 --basis cc-pvtz --world $w --rank $((w/2)) --repeat 5 2>/dev/null | tail -1 >> gpurun_out/shard_h2o32.jsonl
done
cat gpurun_out/shard_h2o32.jsonl
