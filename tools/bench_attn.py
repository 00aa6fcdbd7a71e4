#!/usr/bin/env python
"""Micro-benchmark of keep_attention on the GMFlow swin-window shape (bf16 q/k/v, mode 2), for ablations / rocprofv3.
   python tools/bench_attn.py [n_img] [shift]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import ops  # noqa: E402

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 76
shift = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt = torch.bfloat16 if not (os.environ.get('F32') or os.environ.get('X3')) else torch.float32
if os.environ.get('X3'):
    from comfyui_keep_amd.engine import hiplib as L
    ops.DEFAULT.attn_mma = L.MMA_X3
C, h8, w8 = 128, 64, 64
Ltok = h8 * w8
qkv = torch.randn(n_img * Ltok, 3 * C, device='cuda').to(dt)
o = torch.empty(n_img * Ltok, C, device='cuda')
s = (Ltok * 3 * C, 3 * C, 0)


def run():
    ops.attention(qkv, ops.offset(qkv, C), ops.offset(qkv, 2 * C), o, B=n_img * 4, H=1, Lq=Ltok // 4, Lk=Ltok // 4, D=C, Dv=C,
                  scale=C ** -0.5, q_str=s, k_str=s, v_str=s, o_str=(Ltok * C, C, 0), mode=2, img_h=h8, img_w=w8, ksplit=2,
                  shift=shift, kv_rot=n_img // 2, n_img=n_img)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 4.0 * n_img * 4 * (Ltok // 4) ** 2 * C
print(f"swin attention n_img={n_img} shift={shift} {dt}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s")
if os.environ.get('CHECK'):      # compare with the exact-f32 kernel on the same inputs
    from comfyui_keep_amd.engine import hiplib as L2
    got = o.clone()
    ops.DEFAULT.attn_mma = L2.MMA_F32
    run()
    torch.cuda.synchronize()
    d = (got - o).abs()
    print(f'max |x3 - f32| = {float(d.max()):.3e} (scale {float(o.abs().max()):.3g}); rows off by > 1e-3: '
          f'{int((d.amax(1) > 1e-3).sum())} of {o.shape[0]}')
