#!/usr/bin/env python
"""One pass of the hot path (B clips x T=20 x 512x512, synthetic weights) with nothing else around it -- the target of
rocprofv3 PMC runs (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`).  python tools/run_step.py [bf16|fp32] [B] [passes]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import synth  # noqa: E402
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH  # noqa: E402
from comfyui_keep_amd.engine.net import KeepNet  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
net = KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision(prec)
x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
for _ in range(passes):
    out = net(x)
torch.cuda.synchronize()
print('ok', tuple(out.shape), float(out.abs().mean()))
