cd /root/repo
timeout 900 python -m pytest tests/test_gpu_facelib.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python - <<'PY' 2>&1 | tail -12
import json, sys
sys.path.insert(0, '/root/repo')
import bench
r = bench.facelib_leg()
for k, v in r.items():
    print(k, json.dumps({a: b for a, b in v.items() if a != 'what'}))
PY
