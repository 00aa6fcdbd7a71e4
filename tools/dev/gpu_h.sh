#!/bin/bash
cd /root/repo
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, torch
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import ops, hiplib as L
arena = torch.zeros(1024, device='cuda')
for shape, C, ld in (((16, 4096, 2048), 2048, 2048), ((16, 4096, 2048), 1024, 2048), ((16, 4096, 512), 512, 512), ((16, 4096, 512), 256, 512), ((16, 65536, 128), 128, 128)):
    x = torch.randn(*shape, device='cuda')
    N, R, W = shape
    out = arena[:N]
    for _ in range(3): L.call('keep_absmax', x, out, N, R, C, ld, R * W, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.call('keep_absmax', x, out, N, R, C, ld, R * W, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(shape, 'C', C, f'{ms*1e3:.1f} us  {N*R*C*4/ms/1e6:.0f} GB/s')
PY
