#!/bin/bash
cd /root/repo
{
X3=1 ACT=gelu timeout 300 python tools/bench_conv.py lin256_1024 2>&1 | grep -v amdgpu.ids
X3=1 timeout 300 python tools/bench_conv.py lin256_1024 lin1024_128 lin128 c128_64_1x1 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "x3 or linear or plan" 2>&1 | tail -2
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
