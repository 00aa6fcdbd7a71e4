#!/bin/bash
cd /root/repo
{
for kb in 32 64; do echo "== KB $kb"; KEEP_GATHER_KB=$kb X3=1 timeout 300 python tools/bench_conv.py t512_1024 t512_512 t1024_512 t512_1536 down64_512 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
