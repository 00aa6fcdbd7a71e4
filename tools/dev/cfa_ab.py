"""dev: ms per call at B clips in flight (hipGraph replay) with the CFA block's fused range maxima on / off, alternating in one process.
   python tools/dev/cfa_ab.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import net as net_mod, synth  # noqa: E402
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = net_mod.KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
ref = None
for rnd in range(3):
    for name, free in (('fused', True), ('probed', False)):
        net_mod.CFA_FREE_RANGES = free
        net._graphs = {}
        for _ in range(3):
            out = net(x)
        torch.cuda.synchronize()
        n = 6 if B == 1 else 3
        t0 = time.perf_counter()
        for _ in range(n):
            out = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ref = out.clone() if ref is None else ref
        print(f'B={B} round {rnd} {name:7s} {dt * 1e3:8.2f} ms per call  {B * 20 / dt:7.1f} frames/s  max|diff| vs first: '
              f'{(out - ref).abs().max().item():.3e}', flush=True)
