mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests/test_gpu_net.py -m gpu -q -rf -s -k "T20" 2>&1 | grep -v "^  File\|^Extension" | grep "T20\|T=20\|passed\|failed" | cut -c1-900
timeout 900 python bench.py > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; tail -3 gpurun_out/r2i/bench.err; python -c "
import json;d=json.load(open('gpurun_out/r2i/bench.json'))
print({k:v for k,v in d.items() if k not in ('roofline','other_policies','config','configs')})
r=d['roofline'];print({k:v for k,v in r.items() if k!='all_conv_kernels'})
print({k:(v['value'],v.get('vs_exact_f32_policy')) for k,v in d.get('other_policies',{}).items()})
print(d.get('configs'))
"
