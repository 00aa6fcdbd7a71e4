#!/bin/bash
cd /root/repo
for b in 24 32; do timeout 900 python bench.py --clips $b --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B', $b, d['value'], d['ms_per_step'], d['peak_hbm_gb'])"; done > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
