cd /root/repo
timeout 900 python -m pytest tests/test_gpu_facelib.py -x -q -m gpu -k "parsenet or reflect" 2>&1 | tail -4
timeout 600 python - <<'PY' 2>&1 | tail -12
import json, sys, time
sys.path.insert(0, '/root/repo')
import torch
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import parsenet as PN
eng = PN.ParseNetEngine(PN.synth_parsenet_state_dict(seed=0)).to('cuda')
x = torch.rand((16, 512, 512, 3), device='cuda') * 2 - 1
for n in (16, 3, 1):
    xs = x[:n].contiguous()
    for _ in range(2): eng.classes(xs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.classes(xs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'parsenet {n} faces: {dt * 1e3:.2f} ms per call, {n / dt:.1f} faces/s')
PY
