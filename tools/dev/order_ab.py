"""A/B of one environment switch of the library on the x3 halo layers (AB_ENV, default KEEP_X3_NO_WDMA), alternating in one
   process.  python tools/dev/order_ab.py [N ...]   (written for the work queue of DESIGN 5.3c, which was removed again)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L  # noqa: E402
from comfyui_keep_amd.engine import ops  # noqa: E402

AB_ENV = os.environ.get('AB_ENV', 'KEEP_X3_NO_WDMA')
SHAPES = [(512, 64, 64), (256, 128, 128), (128, 256, 256), (64, 256, 256), (32, 512, 512)]


def time_one(x, w, b, kw, iters):
    for _ in range(2):
        ops.conv(x, w, b, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv(x, w, b, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N in [int(a) for a in sys.argv[1:]] or [4, 16]:
    for hw, cin, cout in SHAPES:
        x = torch.randn(N, hw, hw, cin, device='cuda')
        w = torch.randn(cout, 3, 3, cin, device='cuda') * 0.05
        b = torch.randn(cout, device='cuda')
        sc = ops.x3_scale_for(float(w.abs().max()))
        kw = dict(pad=1, ksize=3, mma=L.MMA_X3, stats=True, wx3=ops.split_x3(w.reshape(-1, cin), sc).view(-1), x3_acc_scale=1.0 / sc,
                  pro=(torch.ones(N, cin, device='cuda'), torch.zeros(N, cin, device='cuda')), pro_act=L.PRO_SWISH)
        iters = max(5, int(40000 / (N * hw * hw * cin * cout / 6.7e7)))
        t = {0: [], 1: []}
        for rep in range(4):
            for so in (0, 1):
                if so:
                    os.environ[AB_ENV] = '1'
                else:
                    os.environ.pop(AB_ENV, None)
                t[so].append(time_one(x, w, b, kw, iters))
        os.environ.pop(AB_ENV, None)
        q, s = min(t[0]), min(t[1])
        print(f'N={N:2d} {cin:3d}->{cout:3d} @{hw:3d}^2  default {q:8.1f} us  {AB_ENV}=1 {s:8.1f} us  ratio {s / q:.3f}   (all: {[round(v) for v in t[0]]} vs {[round(v) for v in t[1]]})', flush=True)
