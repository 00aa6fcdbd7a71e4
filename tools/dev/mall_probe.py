#!/usr/bin/env python
"""Does a depth-first execution order (a few images through several consecutive 512x512 layers before the next images) keep
the intermediate tensors in the 256 MB Infinity Cache?  Chain of L GroupNorm-swish 64->64 3x3 convolutions (x3 policy, fused
statistics) over N images, layer by layer with all N images per launch vs sub-batches of S images through the whole chain."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L  # noqa: E402
from comfyui_keep_amd.engine import ops  # noqa: E402


def main():
    N, H, C, Lyr = 16, 512, int(os.environ.get('C', '64')), 4
    H = int(os.environ.get('H', H))
    torch.manual_seed(0)
    x0 = torch.randn(N, H, H, C, device='cuda')
    ws = [torch.randn(C, 3, 3, C, device='cuda') * 0.04 for _ in range(Lyr)]
    bs = [torch.randn(C, device='cuda') * 0.1 for _ in range(Lyr)]
    g = torch.ones(C, device='cuda')
    be = torch.zeros(C, device='cuda')
    sc = ops.x3_scale_for(max(float(w.abs().max()) for w in ws))
    wx = [ops.split_x3(w.reshape(-1, C), sc).view(-1) for w in ws]
    o = ops.Ops()

    def chain(x):
        st = None
        for l in range(Lyr):
            pro = ops.norm_affine(x, g, be, 32, 1e-6, stats=st)
            x, st = o.conv(x, ws[l], bs[l], pro=pro, pro_act=L.PRO_SWISH, stats=True, mma=L.MMA_X3, wx3=wx[l], x3_acc_scale=1.0 / sc)
        return x

    def run(S):
        outs = [chain(x0[s:s + S]) for s in range(0, N, S)]
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]
    ref = None
    for S in (16, 8, 4, 2, 1):
        for _ in range(2):
            y = run(S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = run(S)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        fl = 2.0 * N * H * H * C * C * 9 * Lyr
        same = True if ref is None else bool(torch.equal(ref, y))
        ref = y if ref is None else ref
        print(f'C={C} H={H} sub-batch {S:2d}: {dt * 1e3:8.2f} ms  {fl / dt / 1e12:6.1f} TF  equal to layer-by-layer: {same}', flush=True)


if __name__ == '__main__':
    main()
