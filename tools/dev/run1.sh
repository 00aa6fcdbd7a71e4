python tools/dev/det_prof.py x3 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/bench_e2e.json 2> gpurun_out/bench_e2e.err; echo rc $?; tail -3 gpurun_out/bench_e2e.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_e2e.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'].get('frac'), d.get('b1',{}).get('value'))
print(json.dumps(d.get('end_to_end'), indent=1)[:1500])
print(d.get('facelib'))
PY
