#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03l; mkdir -p $O
echo "--- worker on the fresh box"; ( time timeout 600 python tests/_native_abi_worker.py ) 2>&1 | grep -E "first PAMD|real|NATIVE"
echo "--- after a process that used 200 GB"; python -c "
import torch; x=torch.empty(25<<30, dtype=torch.float64, device='cuda'); x.fill_(1.0); torch.cuda.synchronize(); print('filled', x.numel()*8/1e9)"
( time timeout 600 python tests/_native_abi_worker.py ) 2>&1 | grep -E "first PAMD|real|NATIVE"
echo "--- while another process holds 150 GB"; python -c "
import torch, time; x=torch.empty(19<<30, dtype=torch.float64, device='cuda'); x.fill_(1.0); torch.cuda.synchronize(); time.sleep(100)" &
sleep 20
( time timeout 600 python tests/_native_abi_worker.py ) 2>&1 | grep -E "first PAMD|real|NATIVE"
wait
