#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
KEEP_X3_EXP=9 X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 2>&1 | grep -v amdgpu.ids | awk '/timeline/ {n++; if (n%23==0) print; next} {next}'
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
