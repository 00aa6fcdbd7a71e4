cd /root/repo
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r5p_bench.json 2> gpurun_out/r5p_bench.err; echo rc=$?
grep '^{' gpurun_out/r5p_bench.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['config']['clips_per_gpu'], d['peak_hbm_gb'], d['roofline']['frac'], d['roofline']['conv_path_frac'], d['roofline'].get('traffic'), d['roofline'].get('algorithmic_bytes_per_launch'), d['roofline'].get('avg_launch_ms'))
print(d.get('clips16',{}).get('value'), d['b1']['value'], d['b1'].get('latency_profile',{}).get('value'), d['cpu_baseline']['value'])
for k, v in d.get('facelib', {}).items():
    print(k, {a: b for a, b in v.items() if a != 'what'})
for leg in ('end_to_end', 'end_to_end_product'):
    for k, v in d.get(leg, {}).items():
        print(leg, k, v.get('value'), v.get('seconds'))
"
