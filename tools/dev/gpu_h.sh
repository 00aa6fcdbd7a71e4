#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
KEEP_DIST_BACKEND=gloo KEEP_DIST_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --clips 4 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-1500
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
