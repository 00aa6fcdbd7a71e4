#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
X3=1 timeout 300 python tools/bench_conv.py c128_64_1x1 lin128 c128_256 c64_512 c256_64 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
