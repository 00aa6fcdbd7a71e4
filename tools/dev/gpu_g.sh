#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/dev/stats_dbg.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stats_dbg.log
cat gpurun_out/stats_dbg.log
