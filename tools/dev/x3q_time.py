"""dev: us per launch of the 64-pixel-block convolution kernels on the shapes of one clip (hipGraph replay); KEEP_HIP_LIB selects an ablation build."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, ops  # noqa: E402

torch.manual_seed(0)
for hw, cin, cout, sk in [(64, 256, 256, 1), (32, 256, 256, 1), (32, 512, 256, 1), (16, 512, 512, None)]:
    w = (torch.randn(cout, 3, 3, cin) * 0.05).cuda()
    sc = ops.x3_scale_for(float(w.abs().max()))
    wx3 = ops.split_x3(w.reshape(-1, cin), sc).view(-1)
    x = torch.randn(1, hw, hw, cin).cuda()
    pro = (torch.rand(1, cin).cuda() + 0.5, torch.randn(1, cin).cuda() * 0.1)
    res = []
    for fl in (0, L.CONV_NO_SMALL_PARTIALS):
        ops.DEFAULT.flags = fl
        kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=1.0 / sc, pro=pro, pro_act=L.PRO_SWISH, split_k=sk, stats=sk == 1)
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
    print(f'{os.environ.get("KEEP_HIP_LIB", "product")[-12:]:12s} {hw:3d}^2 {cin:4d}->{cout:4d}  small blocks {res[0]:6.1f} us   256-pixel kernels {res[1]:6.1f} us', flush=True)
