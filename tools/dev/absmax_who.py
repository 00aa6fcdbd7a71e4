import os, sys, collections, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import ops, synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
os.environ['KEEP_AMD_GRAPH'] = '0'
net = KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision('x3')
cnt = collections.Counter(); byt = collections.Counter()
orig = ops.absmax
def who(x, N, R, C, ld, img_stride, o=None):
    fr = [f for f in traceback.extract_stack() if f.filename.endswith('net.py')]
    key = ' <- '.join(f'{f.name}:{f.lineno}' for f in fr[-2:])
    cnt[key] += 1; byt[key] += N * R * C * 4
    return orig(x, N, R, C, ld, img_stride, o)
ops.absmax = who
x = synth.synth_clip(T=20, B=4, seed=1234).cuda()
net(x); torch.cuda.synchronize()
for k, v in sorted(byt.items(), key=lambda kv: -kv[1])[:25]:
    print(f'{v / 1e9:8.2f} GB  n={cnt[k]:4d}  {k}')
