#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/stab.log
cat gpurun_out/stab.log
timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
