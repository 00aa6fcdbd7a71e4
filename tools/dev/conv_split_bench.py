"""dev: us per 3x3 convolution (kernel + split-K reducer, hipGraph replay of 20 launches) on the small maps of the frame recurrence,
by forced split_k and images per launch.   python tools/dev/conv_split_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, ops  # noqa: E402

SHAPES = [(16, 512, 512), (32, 256, 256), (64, 256, 256), (32, 512, 256)]
FLAGS = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
ops.DEFAULT.flags = FLAGS
print('flags', hex(FLAGS))
torch.manual_seed(0)
for hw, cin, cout in SHAPES:
    w = (torch.randn(cout, 3, 3, cin) * 0.05).cuda()
    sc = ops.x3_scale_for(float(w.abs().max()))
    wx3 = ops.split_x3(w.reshape(-1, cin), sc).view(-1)
    for n_img in (1, 16):
        x = torch.randn(n_img, hw, hw, cin).cuda()
        pro = (torch.rand(n_img, cin).cuda() + 0.5, torch.randn(n_img, cin).cuda() * 0.1)
        res = []
        for sk in (1, 2, 4, 8, 16, 32):
            ops.DEFAULT.flags = FLAGS
            if sk > cin // 32:
                res.append(float('nan'))
                continue
            kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=1.0 / sc, pro=pro, pro_act=L.PRO_SWISH, split_k=sk)
            for _ in range(3):
                y = ops.conv(x, w, None, **kw)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    y = ops.conv(x, w, None, **kw)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 20)
        print(f'{hw:4d}^2 {cin:4d}->{cout:4d} | {n_img:2d} images | split 1/2/4/8/16/32: ' + ' '.join(f'{r:7.1f}' for r in res), flush=True)
