#!/bin/bash
cd /root/repo
{
for w in "" 1; do echo "== WIDE=$w"; env ${w:+KEEP_GATHER_WIDE=1} X3=1 ACT=gelu timeout 300 python tools/bench_conv.py lin256_1024 2>&1 | grep -v amdgpu.ids | cut -c1-150; env ${w:+KEEP_GATHER_WIDE=1} X3=1 timeout 300 python tools/bench_conv.py lin256_1024 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
