#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/shard_probe.py --nwater 128 --basis cc-pvdz --world 8 --rank 3 > gpurun_out/shard_h2o128.json 2> gpurun_out/shard_h2o128.err
tail -2 gpurun_out/shard_h2o128.err; cat gpurun_out/shard_h2o128.json
