timeout 900 python -m pytest tests -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bf16', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']); f=d['fp32_parity_policy']; print('fp32', f['value'], f['ms_per_step'], f['roofline']['achieved'], f['roofline']['frac'], f['roofline']['conv_path_tflops'])"
