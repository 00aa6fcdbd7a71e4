#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
echo "== nb 15"; timeout 900 python tools/dev/batch_flip2.py 15 2>&1 | grep -v amdgpu.ids | grep -v "max diff [01]," 
echo "== nb 15 window attn f32"; KEEP_ATTN_DBG=128 timeout 900 python tools/dev/batch_flip2.py 15 2>&1 | grep -v amdgpu.ids | grep -v "max diff [01],"
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
