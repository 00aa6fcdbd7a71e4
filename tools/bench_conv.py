#!/usr/bin/env python
"""Micro-benchmark of keep_conv2d on the hot layer shapes (through the C-ABI), for rocprofv3 / PMC runs.
   python tools/bench_conv.py [layer ...]   layers: c64_512 c128_256 up128_512 c256_64 c512_16 lin128"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L  # noqa: E402
if os.environ.get('ABL_LIB'):      # dev library (-DKEEP_X3_ABLATE: tools/dev/README.md)
    L.LIB_PATH = os.environ['ABL_LIB']
from comfyui_keep_amd.engine import ops  # noqa: E402
if os.environ.get('CONV_FLAGS'):      # keep_conv2d_args.flags overrides (hiplib.CONV_*), e.g. 512 = KEEP_CONV_NO_STREAM
    ops.DEFAULT.flags = int(os.environ['CONV_FLAGS'])

LAYERS = {  # name: (N, H, W, Cin, Cout, ksize, upsample)
    'c64_512': (4, 512, 512, 64, 64, 3, False),
    'c128_256': (4, 256, 256, 128, 128, 3, False),
    'up128_512': (4, 256, 256, 128, 128, 3, True),
    'c256_64': (4, 64, 64, 256, 256, 3, False),
    'c512_16': (4, 16, 16, 512, 512, 3, False),
    'c64_512_n48': (48, 512, 512, 64, 64, 3, False),        # the frame loop's launches at the bench's 48 clips per call
    'c128_256_n48': (48, 256, 256, 128, 128, 3, False),
    'c64_512_n1': (1, 512, 512, 64, 64, 3, False),          # one clip in flight: the streaming kernel on ONE image
    'c128_256_n1': (1, 256, 256, 128, 128, 3, False),
    'c256_64_n1': (1, 64, 64, 256, 256, 3, False),          # one clip in flight: the 64-pixel-block kernels (conv3x3_x3q_kernel)
    'c256_32_n1': (1, 32, 32, 256, 256, 3, False),
    'lin128': (1, 622592, 1, 128, 128, 1, False),
    'lin256_1024': (1, 622592, 1, 256, 1024, 1, False),
    't512_1024': (1, 4096, 1, 512, 1024, 1, False),
    't512_512': (1, 4096, 1, 512, 512, 1, False),
    't1024_512': (1, 4096, 1, 1024, 512, 1, False),
    't512_1536': (16, 256, 1, 512, 1536, 1, False),
    'lin1024_128': (1, 622592, 1, 1024, 128, 1, False),
    'c128_64_1x1': (4, 512, 512, 128, 64, 1, False),
    'down64_512': (4, 512, 512, 64, 64, 3, False),
    'c512_16_b8': (8, 16, 16, 512, 512, 3, False),
    'lin512_1024': (1, 1024, 1, 512, 1024, 1, False),
    'lin1024_512': (1, 1024, 1, 1024, 512, 1, False),
    'lin512_b8': (1, 2048, 1, 512, 512, 1, False),
    'lin1024_b8': (1, 2048, 1, 1024, 512, 1, False),
}


def run(name, mma, in_bf16, iters=int(os.environ.get('ITERS', '20'))):
    N, H, W, Cin, Cout, k, up = LAYERS[name]
    x = torch.randn(N, H, W, Cin, device='cuda')
    w = torch.randn(Cout, k, k, Cin, device='cuda') * 0.05
    b = torch.randn(Cout, device='cuda')
    wb = w.to(torch.bfloat16)
    pro = None
    if in_bf16:
        pro = (torch.ones(N, Cin, device='cuda'), torch.zeros(N, Cin, device='cuda'))
    kw = dict(pad=k // 2, ksize=k, upsample=up, mma=mma, wb=wb, stats=not os.environ.get('NOSTATS'))      # NOSTATS=1: plain linear launches
    if os.environ.get('NOSTATS'):
        kw['bounded'] = True
    if mma == L.MMA_X3:
        sc = ops.x3_scale_for(float(w.abs().max()))
        kw.update(wx3=ops.split_x3(w.reshape(-1, Cin), sc).view(-1), x3_acc_scale=1.0 / sc)
    if os.environ.get('RES'):
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        kw['residual'] = torch.randn(N, Ho, Wo, Cout, device='cuda')
    if os.environ.get('ACT') == 'gelu':
        kw['act'] = L.ACT_GELU
    if os.environ.get('UP2') and up and mma == L.MMA_X3:      # nearest x2 + 3x3 as four 2x2-tap phase convolutions
        w4 = ops.up2_phase_weights(w)
        sc4 = ops.x3_scale_for(float(w4.abs().max()))
        kw.update(upsample=L.UPSAMPLE_X2_PHASES, wx3=ops.split_x3(w4.reshape(-1, Cin), sc4).view(-1), x3_acc_scale=1.0 / sc4)
    if os.environ.get('DOWN'):       # VQGAN Downsample geometry: 3x3 stride 2, pad right / bottom only
        kw.update(down=True, pad=0)
    if os.environ.get('SPLITK'):
        kw['split_k'] = int(os.environ['SPLITK'])
    if pro is not None:
        kw.update(pro=pro, pro_act=L.PRO_SWISH)
    for _ in range(3):
        y = ops.conv(x, w, b, **kw)
    if isinstance(y, tuple):
        y = y[0]
    torch.cuda.synchronize()
    ops.DEFAULT.profile = []
    for _ in range(iters):
        y = ops.conv(x, w, b, **kw)
    torch.cuda.synchronize()
    rec, ops.DEFAULT.profile = ops.DEFAULT.profile, None
    ms = sum(r[3].elapsed_time(r[4]) for r in rec) / len(rec)
    fl = rec[0][1]
    print(f"{name:10s} mma={('f32 ', 'bf16', 'x3  ')[mma]} pre-activated-bf16-input={in_bf16!s:5s} kernel={rec[0][0]:22s} "
          f"{ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s (incl. the norm_act pass when present)")
    return y


if __name__ == '__main__':
    names = sys.argv[1:] or list(LAYERS)
    if os.environ.get('F32'):
        for n in names:
            run(n, L.MMA_F32, False)
        sys.exit(0)
    if os.environ.get('X3'):      # split fp16: plain input, and with the fused GroupNorm affine + swish prologue
        for n in names:
            if not os.environ.get('PRO_ONLY'):
                run(n, L.MMA_X3, False)
            if LAYERS[n][5] == 3:
                run(n, L.MMA_X3, True)
        sys.exit(0)
    for n in names:
        run(n, L.MMA_BF16, False)
        if LAYERS[n][5] == 3:
            run(n, L.MMA_BF16, True)
