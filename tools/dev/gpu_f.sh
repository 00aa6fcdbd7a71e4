mkdir -p gpurun_out/r2l
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^  File\|^Extension" | tail -12 > gpurun_out/r2l/tests.log
tail -8 gpurun_out/r2l/tests.log
bash tools/profile_step.sh x3 16 r2l > gpurun_out/r2l/profile_x3_16.out 2>&1; tail -26 gpurun_out/r2l/profile_x3_16.out
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err; tail -3 gpurun_out/r2l/bench.err; python -c "
import json;d=json.load(open('gpurun_out/r2l/bench.json'))
print({k:v for k,v in d.items() if k not in ('roofline','other_policies','config','configs')})
r=d['roofline'];print({k:v for k,v in r.items() if k!='all_conv_kernels'})
print({k:(v['value'],v.get('vs_exact_f32_policy')) for k,v in d.get('other_policies',{}).items()})
print(d.get('configs'))
"
