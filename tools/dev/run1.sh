cd /root/repo
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "gmflow" 2>&1 | grep -v "^$" | cut -c1-300 | tail -6
for v in 0 8 4 16; do echo "== KEEP_GM_FFN_IMAGES=$v"; KEEP_GM_FFN_IMAGES=$v python bench.py --no-extras --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
