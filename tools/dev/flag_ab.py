"""dev: ms per call at B clips in flight (hipGraph replay) with keep_conv2d flag sets toggled in one process, alternating.
   python tools/dev/flag_ab.py [B] [name=flagbits ...]      (default: new=0 old=CONV_NO_GEMM_LAT)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, net as net_mod, synth  # noqa: E402
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sets = [a.split('=') for a in sys.argv[2:]] or [('new', '0'), ('old', str(L.CONV_NO_GEMM_LAT))]
sets = [(n, int(v, 0)) for n, v in sets]
net = net_mod.KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
outs = {}
for rnd in range(3):
    for name, fl in sets:
        net.o.flags = fl
        for _ in range(2 if rnd else 3):
            out = net(x)
        torch.cuda.synchronize()
        n = 5 if B == 1 else 3
        t0 = time.perf_counter()
        for _ in range(n):
            out = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        outs.setdefault(name, out.clone())
        print(f'B={B} round {rnd} {name:8s} flags {fl:#x} {dt * 1e3:8.2f} ms per call  {B * 20 / dt:7.1f} frames/s  '
              f'max|diff| vs {sets[0][0]}: {(out - outs[sets[0][0]]).abs().max().item():.3e}', flush=True)
