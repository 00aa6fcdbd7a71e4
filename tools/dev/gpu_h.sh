#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
for m in 4096 2048; do
echo "== KEEP_GATHER_SMALL_M=$m"
KEEP_GATHER_SMALL_M=$m timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in d['roofline']['all_conv_kernels'].items() if 'conv_x3_kernel' in k]"
done
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
