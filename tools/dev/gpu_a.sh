mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests -m gpu -q -rf -x 2>&1 | grep -v "^  File\|^Extension" | tail -40 > gpurun_out/r2h/tests.log
tail -30 gpurun_out/r2h/tests.log
timeout 900 python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; tail -3 gpurun_out/r2h/bench.err; python -c "
import json;d=json.load(open('gpurun_out/r2h/bench.json'))
print({k:v for k,v in d.items() if k not in ('roofline','other_policies','config','configs')})
r=d['roofline'];print({k:v for k,v in r.items() if k!='all_conv_kernels'})
print({k:(v['value'],v.get('vs_exact_f32_policy')) for k,v in d.get('other_policies',{}).items()})
print(d.get('configs'))
"
