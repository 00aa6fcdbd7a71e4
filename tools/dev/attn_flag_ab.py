"""dev: ms per call at B clips in flight with keep_attention flag sets toggled in one process.  python tools/dev/attn_flag_ab.py [B] [name=bits ...]"""
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
sets = [a.split('=') for a in sys.argv[2:]] or [('new', '0'), ('old', str(L.ATTN_NO_SMALL))]
sets = [(n, int(v, 0)) for n, v in sets]
net = net_mod.KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
outs = {}
for rnd in range(3):
    for name, fl in sets:
        net.o.attn_flags = fl
        net._graphs.clear(); net._graph_seen.clear()
        for _ in range(3):
            out = net(x)
        torch.cuda.synchronize()
        n = 5 if B == 1 else 3
        t0 = time.perf_counter()
        for _ in range(n):
            out = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        outs.setdefault(name, out.clone())
        print(f'B={B} round {rnd} {name:8s} attn flags {fl:#x} {dt * 1e3:8.2f} ms per call  {B * 20 / dt:7.1f} frames/s  '
              f'max|diff| vs {sets[0][0]}: {(out - outs[sets[0][0]]).abs().max().item():.3e}', flush=True)
