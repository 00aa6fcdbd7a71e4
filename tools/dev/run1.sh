python bench.py > gpurun_out/r4p_bench.json 2> gpurun_out/r4p_bench.err; echo rc $?
KEEP_DIST_DEVICE=0 python bench.py --gpus 2 --clips 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4p_bench_2ranks_1gpu.json 2> gpurun_out/r4p_bench2.err; echo rc $?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4p_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'frac', d['roofline'].get('frac'), 'conv_path', d['roofline'].get('conv_path_frac'), 'b1', d.get('b1',{}).get('value'), 'pcie', d.get('pcie_inclusive',{}).get('value'))
print({k:v['value'] for k,v in d['configs'].items()}, {k:v['value'] for k,v in d['end_to_end'].items()})
print(d['cpu_baseline'])
print(d['roofline'])
d2=json.loads([l for l in open('gpurun_out/r4p_bench_2ranks_1gpu.json') if l.startswith('{')][-1])
print('2 ranks on 1 gpu:', d2['value'], d2.get('broadcast_ms'), d2.get('config5_one_video_per_gpu',{}).get('value'))
PY
