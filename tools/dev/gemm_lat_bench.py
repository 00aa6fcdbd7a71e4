"""dev: us per launch of the token GEMMs of the frame recurrence, per kernel form and images per launch.
   python tools/dev/gemm_lat_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, ops  # noqa: E402

SHAPES = [(256, 512, 512), (256, 512, 1024), (256, 1024, 512), (256, 512, 1536), (256, 256, 512), (256, 2048, 512), (256, 512, 4096),
          (1024, 1024, 256), (1024, 512, 256), (1024, 256, 2048), (1024, 256, 1024)]
FORMS = [('seq', L.CONV_NO_GEMM_LAT), ('waves', L.CONV_GEMM_LAT_WAVES), ('tiles', L.CONV_GEMM_LAT_TILES)]
torch.manual_seed(0)
print('hw K N | images: ' + ' '.join(f'{f:>7s}' for f, _ in FORMS) + ' (us per launch)')
for hw, K, N in SHAPES:
    w = (torch.randn(N, K) * 0.05).cuda()
    sc = ops.x3_scale_for(float(w.abs().max()))
    wx3 = ops.split_x3(w, sc).view(-1)
    for n_img in (1, 2, 4, 8, 16):
        x = torch.randn(n_img, hw, 1, K).cuda()
        res = []
        for name, fl in FORMS:
            ops.DEFAULT.flags = fl
            kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=1.0 / sc, pad=0, ksize=1, bounded=True)
            for _ in range(3):
                y = ops.conv(x, w, None, **kw)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()              # 40 dependent-in-stream launches replayed as one graph: kernel time, not host time
            with torch.cuda.graph(g):
                for _ in range(40):
                    y = ops.conv(x, w, None, **kw)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 40)
        print(f'{hw:5d} {K:5d} {N:5d} | {n_img:2d}: ' + ' '.join(f'{r:7.1f}' for r in res), flush=True)
