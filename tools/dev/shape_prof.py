"""Per-(kernel, shape) time table of the conv launches of one pass.  python tools/dev/shape_prof.py [prec] [B]"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
prec = sys.argv[1] if len(sys.argv) > 1 else 'x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
net = KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision(prec)
x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
net(x); torch.cuda.synchronize()
net.o.profile = []
net(x); torch.cuda.synchronize()
rec, net.o.profile = net.o.profile, None
agg = collections.OrderedDict()
for r in rec:
    k = (r[0], r[6], r[2])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += r[3].elapsed_time(r[4]); a[2] += r[1]
tot = sum(a[1] for a in agg.values())
print(f'conv launches {len(rec)}  total {tot:.1f} ms')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{a[1]:8.2f} ms  n={a[0]:4d}  {a[2]/a[1]/1e9:7.1f} TF  {k[0]:42s} NHW,Ci,Co,k,s,up,pro={k[1]} split={k[2]}')
