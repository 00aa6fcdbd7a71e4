#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
timeout 300 python tools/dev/cft_dbg.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
