"""dev: ParseNet(512) on 16 faces, a few calls (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import parsenet as PN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
par = PN.ParseNetEngine(PN.synth_parsenet_state_dict(seed=0)).to('cuda')
x = torch.rand((n, 512, 512, 3), device='cuda') * 2 - 1
for _ in range(4):
    par.classes(x)
torch.cuda.synchronize()
