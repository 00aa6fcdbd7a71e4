#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k "x3 and (T3 or stacks or hipgraph or u8)" 2>&1 | tail -3
timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k,v) for k,v in d['roofline']['all_conv_kernels'].items()]"
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
