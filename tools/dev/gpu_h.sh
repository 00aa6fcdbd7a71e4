#!/bin/bash
cd /root/repo
{
for e in 0 10; do echo "== EXP $e"; KEEP_X3_EXP=$e X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 c256_64 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
