#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gm_mlp" 2>&1 | grep -v "^  File" | tail -12
timeout 600 python - <<'PY'
import os, sys, torch
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L, ops
C, M = 128, 4096 * 152
a, b = torch.randn(M, C, device='cuda'), torch.randn(M, C, device='cuda')
w0, w2 = torch.randn(8 * C, 2 * C, device='cuda') * 0.06, torch.randn(C, 8 * C, device='cuda') * 0.03
sc = ops.x3_scale_for(float(max(w0.abs().max(), w2.abs().max())))
w0x, w2x = ops.split_x3(w0, sc).view(-1), ops.split_x3(w2, sc).view(-1)
out = torch.empty(M, C, device='cuda')
for env in (None, '1'):
    if env: os.environ['KEEP_MLPX_HC64'] = env
    for _ in range(2): L.call('keep_gm_mlp_x3', a, b, w0x, w2x, out, M, C, 1.0 / sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): L.call('keep_gm_mlp_x3', a, b, w0x, w2x, out, M, C, 1.0 / sc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'keep_gm_mlp_x3 HC={"64" if env else "32"} M={M}: {ms*1e3:.0f} us  {2.0*M*(256*1024+1024*128)/ms/1e9:.1f} TFLOP/s')
PY
} > gpurun_out/exp_h.log 2>&1
cat gpurun_out/exp_h.log
