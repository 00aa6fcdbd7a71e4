#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2b_tests.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
tail -c 1500 gpurun_out/r2b_tests.log; tail -3 gpurun_out/r2b_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r2b_bench.json'))
print({k:v for k,v in d.items() if k not in ('roofline','other_policies','config','configs')})
r=d['roofline'];print({k:v for k,v in r.items() if k!='all_conv_kernels'})
for k,v in r['all_conv_kernels'].items(): print(k, v)
print({k:(v['value'],v.get('vs_exact_f32_policy')) for k,v in d.get('other_policies',{}).items()})
print(d.get('configs'))
"
timeout 600 python tools/dev/shape_prof.py x3 16 2>&1 | grep -v amdgpu > gpurun_out/shape_prof_x3.log
