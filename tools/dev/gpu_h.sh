#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/dev/absmax_who.py 2>&1 | grep -v amdgpu.ids > gpurun_out/exp_h.log
cat gpurun_out/exp_h.log
