#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -2 gpurun_out/bench_default.err
python -c "
import json;d=json.load(open('gpurun_out/bench_default.json'))
print({k:v for k,v in d.items() if k not in ('roofline','other_policies','config','configs')})
"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
