"""dev: does Infinity-Cache residency of the weights matter for the latency-bound launches of the frame recurrence?
us per launch with the weights rotating over N copies: N = 1 (L2-warm), 75 MB (fits the 256 MB Infinity Cache, misses L2), 600 MB (HBM).
   python tools/dev/cold_weights_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, ops  # noqa: E402

torch.manual_seed(0)
CASES = [('conv 16^2 512->512 3x3', (1, 16, 16, 512), (512, 3, 3, 512), dict(pro_act=L.PRO_SWISH)),
         ('conv 32^2 256->256 3x3', (1, 32, 32, 256), (256, 3, 3, 256), dict(pro_act=L.PRO_SWISH)),
         ('gemm 256 x 512 -> 512', (1, 256, 1, 512), (512, 1, 1, 512), dict(pad=0, ksize=1, bounded=True)),
         ('gemm 256 x 1024 -> 512', (1, 256, 1, 1024), (512, 1, 1, 1024), dict(pad=0, ksize=1, bounded=True))]
for name, xs, ws, extra in CASES:
    x = torch.randn(*xs).cuda()
    cin = ws[-1]
    wbytes = ws[0] * ws[1] * ws[2] * ws[3] * 4
    pro = (torch.rand(xs[0], cin).cuda() + 0.5, torch.randn(xs[0], cin).cuda() * 0.1) if 'pro_act' in extra else None
    for total_mb in (0, 75, 600):
        n = max(1, int(total_mb * 1e6 / (2 * wbytes)))          # fp32 tensor + its x3 twin travel together; only the twin is read
        ws_list = []
        for i in range(n):
            w = (torch.randn(*ws) * 0.05).cuda()
            sc = ops.x3_scale_for(float(w.abs().max()))
            ws_list.append((w, ops.split_x3(w.reshape(-1, cin), sc).view(-1), 1.0 / sc))
        reps = max(40, 2 * n)

        def run():
            for i in range(reps):
                w, wx3, asc = ws_list[i % n]
                ops.conv(x, w, None, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pro=pro, **extra)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f'{name:26s} weights rotating over {n:4d} copies ({n * wbytes / 1e6:7.1f} MB of x3 twins): {e0.elapsed_time(e1) * 1e3 / reps:7.1f} us per launch', flush=True)
        del ws_list
