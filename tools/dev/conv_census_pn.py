"""dev: per-(kernel, shape) census of ParseNet(512)'s keep_conv2d launches at 16 faces: time, TF/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import parsenet as PN  # noqa: E402

par = PN.ParseNetEngine(PN.synth_parsenet_state_dict(seed=0)).to('cuda')
x = torch.rand((16, 512, 512, 3), device='cuda') * 2 - 1
for _ in range(2):
    par.classes(x)
torch.cuda.synchronize()
par.o.profile = []
par.classes(x)
torch.cuda.synchronize()
rec, par.o.profile = par.o.profile, None
tot = 0.0
for cfg, flops, split_k, e0, e1, nbytes, shape in rec:
    ms = e0.elapsed_time(e1)
    tot += ms
    print(f'{cfg[:44]:44s} {str(shape):44s} sk{split_k} {ms * 1e3:9.1f} us {flops / ms / 1e9:7.1f} TF {nbytes / ms / 1e6:7.0f} GB/s')
print(f'total {tot:.2f} ms in {len(rec)} launches')
