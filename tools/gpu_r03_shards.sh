#!/bin/bash
# compute side of the scaling curves with the final kernels: one rank's shard alone on one GPU
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03shards; mkdir -p $O
for w in 1 2 4 8; do
  timeout 300 python tools/shard_probe.py --nwater 32 --basis cc-pvtz --world $w --rank $((w/2)) --repeat 5 2>/dev/null | tail -1 | tee -a $O/shard_probe_h2o32_world1_2_4_8.jsonl | cut -c1-330
done
timeout 600 python tools/shard_probe.py --molecule taxol --basis def2-tzvp --world 8 --rank 3 --repeat 5 2>/dev/null | tail -1 | tee $O/shard_probe_taxol_rank3of8.json | cut -c1-400
timeout 600 python tools/shard_probe.py --nwater 128 --basis cc-pvdz --world 8 --rank 3 2>/dev/null | tail -1 | tee $O/shard_probe_h2o128_rank3of8.json | cut -c1-400
