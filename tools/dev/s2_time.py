"""dev: the encoder's stride-2 convolutions with statistics at one image: us per launch (hipGraph replay of 40) with the small tile + statistics
replica (flags 0) and with the plan's 128 x 128 tile (KEEP_CONV_NO_SMALL_PARTIALS)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, ops  # noqa: E402

o = ops.Ops()
o.set_precision(L.MMA_X3) if hasattr(o, 'mma') and False else None
for (n, h, c_in, c_out) in ((1, 64, 256, 256), (1, 32, 256, 256), (1, 128, 128, 128), (1, 256, 128, 128), (1, 512, 64, 64)):
    x = torch.randn((n, h, h, c_in), device='cuda')
    w = torch.randn((c_out, 3, 3, c_in), device='cuda') * 0.05
    b = torch.randn((c_out,), device='cuda')
    sc = ops.x3_scale_for(float(w.abs().max()))
    wx3 = ops.split_x3(w.reshape(-1, c_in), sc).view(-1)
    line = f'({n},{h},{h},{c_in}->{c_out}, s2)'
    for fl in (0, L.CONV_NO_SMALL_PARTIALS):
        o.flags = fl
        o.begin_forward(torch.device('cuda'))
        kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=1.0 / sc, bounded=True, stats=True, down=True)
        for _ in range(3):
            o.conv(x, w, b, **kw)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            o.begin_forward(torch.device('cuda'))
            with torch.cuda.graph(g):
                for _ in range(40):
                    o.conv(x, w, b, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        line += f'   flags {fl:#x}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us'
    print(line, flush=True)
