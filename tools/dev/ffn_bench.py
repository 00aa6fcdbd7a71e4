"""dev: keep_gm_ffn_x3 (one launch) against the two x3 GEMM launches it replaces, at GMFlow's token counts.
   python tools/dev/ffn_bench.py [M ...]      (default 622592 = 4 clips x 38 pair members x 4096, 2490368 = 16 clips)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, ops  # noqa: E402

C, Hd = 128, 1024
Ms = [int(a) for a in sys.argv[1:]] or [622592, 2490368]
torch.manual_seed(0)
w0, w2 = torch.randn(Hd, 2 * C, device='cuda') * 0.08, torch.randn(C, Hd, device='cuda') * 0.05
g, be = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1


def x3w(wp):
    sc = ops.x3_scale_for(float(wp.abs().max()))
    return ops.split_x3(wp.reshape(-1, wp.shape[-1]), sc).view(-1), 1.0 / sc


wx0, a0 = x3w(w0)
wx2, a2 = x3w(w2)
wx2p, a2p = x3w(ops.ffn_w2_perm(w2))
for M in Ms:
    src, msg = torch.randn(M, C, device='cuda'), torch.randn(M, C, device='cuda')
    out = torch.empty(M, C, device='cuda')
    n_img = M // 4096
    kw = dict(mma=L.MMA_X3, pad=0, ksize=1, bounded=True)
    s4, m4 = src.view(n_img, 4096, 1, C), msg.view(n_img, 4096, 1, C)

    def fused():
        L.call('keep_gm_ffn_x3', src, msg, wx0, float(a0), wx2p, float(a2p), g, be, 1e-5, out, M, C, Hd, 0)

    def two():
        hm = ops.conv(s4, w0, None, act=L.ACT_GELU, x2=m4, wx3=wx0, x3_acc_scale=a0, **kw)
        return ops.conv(hm, w2, None, residual=s4, ln=(g, be, 1e-5), wx3=wx2, x3_acc_scale=a2, **kw)

    for name, fn in (('two launches', two), ('fused', fused), ('two launches', two), ('fused', fused)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        fl = 2.0 * M * (2 * C * Hd + Hd * C)
        print(f'M={M:8d} {name:13s} {us:9.1f} us  {fl / us * 1e-6:7.1f} TF algorithmic', flush=True)
    d = (two().reshape(M, C) - out).abs().max().item()
    print(f'M={M}: max |fused - two| = {d:.3e}', flush=True)
