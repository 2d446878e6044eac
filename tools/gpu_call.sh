cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/two_rank_xc_check.py 32 2>&1 | grep "nelec" | tail -2
