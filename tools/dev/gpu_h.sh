#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "vq_nearest" 2>&1 | grep -v "^  File\|amdgpu.ids" | tail -25
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
