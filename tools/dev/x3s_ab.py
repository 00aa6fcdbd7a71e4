#!/usr/bin/env python
"""A/B of the wave-specialised halo kernel (KEEP_X3S=1) against the single-role one (KEEP_X3S=0): results must be bit-identical
(same chunk / tap / MFMA order), timings side by side.   python tools/dev/x3s_ab.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L  # noqa: E402
from comfyui_keep_amd.engine import ops  # noqa: E402

CASES = [  # name, N, H, W, Cin, Cout, upsample, pro, residual, act, split_k
    ('c64_512', 16, 512, 512, 64, 64, False, True, True, 0, 0),
    ('c64_512 nopro', 16, 512, 512, 64, 64, False, False, False, 0, 0),
    ('c128_256', 16, 256, 256, 128, 128, False, True, True, 0, 0),
    ('up128_512', 16, 256, 256, 128, 128, True, False, False, 0, 0),
    ('c128_64@512', 16, 512, 512, 128, 64, False, True, False, 0, 0),
    ('c128_128', 16, 128, 128, 128, 128, False, True, True, 0, 0),
    ('c256_64', 16, 64, 64, 256, 256, False, True, True, 0, 0),
    ('c256_32', 16, 32, 32, 256, 256, False, True, True, 0, 0),
    ('c512_16', 16, 16, 16, 512, 512, False, True, True, 0, 0),
    ('c512_16 split4', 16, 16, 16, 512, 512, False, True, True, 0, 4),
    ('c256_64 lrelu', 4, 64, 64, 256, 512, False, False, False, L.ACT_LRELU02, 0),
    ('c96_128 relu', 8, 128, 128, 96, 96, False, True, False, 0, 0),
    ('c64_512 N=320', 320, 512, 512, 64, 64, False, True, True, 0, 0),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for name, N, H, W, Cin, Cout, up, pro, res, act, sk in CASES:
        if only and only not in name:
            continue
        if N * H * W * max(Cin, Cout) * 4 > 24e9:
            N = 64
        torch.manual_seed(0)
        x = torch.randn(N, H, W, Cin, device='cuda')
        w = torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05
        b = torch.randn(Cout, device='cuda')
        sc = ops.x3_scale_for(float(w.abs().max()))
        kw = dict(upsample=up, mma=L.MMA_X3, wx3=ops.split_x3(w.reshape(-1, Cin), sc).view(-1), x3_acc_scale=1.0 / sc, stats=True, act=act)
        if pro:
            kw.update(pro=(torch.rand(N, Cin, device='cuda') + 0.5, torch.randn(N, Cin, device='cuda') * 0.1), pro_act=L.PRO_SWISH)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        if res:
            kw['residual'] = torch.randn(N, Ho, Wo, Cout, device='cuda')
        if sk:
            kw['split_k'] = sk
        outs, times = {}, {}
        for mode in ('0', '1'):
            os.environ['KEEP_X3S'] = mode
            for _ in range(2):
                y, st = ops.conv(x, w, b, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                y, st = ops.conv(x, w, b, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[mode] = e0.elapsed_time(e1) / iters
            outs[mode] = (y.clone(), None if st is None or st.part is None else st.part.clone(),
                          None if st is None or st.amax is None else st.amax.clone())
        same = torch.equal(outs['0'][0], outs['1'][0])
        same_st = (outs['0'][1] is None and outs['1'][1] is None) or torch.equal(outs['0'][1], outs['1'][1])
        same_am = (outs['0'][2] is None and outs['1'][2] is None) or torch.equal(outs['0'][2], outs['1'][2])
        fl = 2.0 * N * Ho * Wo * Cout * 9 * Cin
        print(f'{name:18s} N={N:3d} single-role {times["0"] * 1e3:8.1f} us {fl / times["0"] / 1e9:6.1f} TF | specialised {times["1"] * 1e3:8.1f} us '
              f'{fl / times["1"] / 1e9:6.1f} TF  x{times["0"] / times["1"]:.3f} | equal out={same} stats={same_st} amax={same_am} '
              f'finite={bool(torch.isfinite(outs["1"][0]).all())}', flush=True)
        if not same:
            d = (outs['0'][0] - outs['1'][0]).abs()
            print('   max diff', float(d.max()), 'at', torch.nonzero(d == d.max())[0].tolist(), 'frac differing', float((d > 0).float().mean()))
    os.environ.pop('KEEP_X3S', None)


if __name__ == '__main__':
    main()
