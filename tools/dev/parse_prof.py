"""ParseNet(512) on the engine, 16 faces per call, 3 calls (dev: run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import parsenet as PN
eng = PN.ParseNetEngine(PN.synth_parsenet_state_dict(seed=0)).to('cuda')
x = torch.rand((16, 512, 512, 3), device='cuda') * 2 - 1
eng.o.profile = [] if len(sys.argv) > 1 else None
for _ in range(3):
    eng.classes(x)
torch.cuda.synchronize()
if eng.o.profile:
    rec = eng.o.profile[len(eng.o.profile) * 2 // 3:]
    tot = 0.0
    for cfg, flops, sk, e0, e1, nbytes, shape in rec:
        ms = e0.elapsed_time(e1); tot += ms
        print(f'{cfg[:46]:46s} {str(shape):40s} sk{sk} {ms * 1e3:8.1f} us {flops / ms / 1e9:7.1f} TF')
    print('conv total', round(tot, 2), 'ms')
