"""Run one forward under the x3 policy and, for every attention call, also run the exact-f32 kernel on the same inputs;
report calls whose outputs differ by more than 1e-3 of the output scale.   python tools/dev/attn_cmp.py [B] [T]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L, ops, synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
os.environ['KEEP_AMD_GRAPH'] = '0'
net = KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision('x3')
orig = ops.Ops.attention
seen = {}
def both(self, q, k, v, o, **kw):
    r = orig(self, q, k, v, o, **kw)
    if kw.get('mma') is None and self.attn_mma == L.MMA_X3 and q.dtype == torch.float32:
        o2 = torch.empty_like(o)
        kw2 = dict(kw); kw2['mma'] = L.MMA_F32; kw2['probe'] = False
        orig(self, q, k, v, o2, **kw2)
        torch.cuda.synchronize()
        d = float((o - o2).abs().max()); sc = float(o2.abs().max())
        key = tuple((k_, kw[k_]) for k_ in ('B', 'H', 'Lq', 'Lk', 'D', 'Dv', 'mode', 'shift', 'kv_rot', 'n_img', 'probe') if k_ in kw)
        bad = d > 1e-3 * max(sc, 1e-6) or not torch.isfinite(o).all()
        if key not in seen or bad:
            seen[key] = True
            print(('BAD ' if bad else 'ok  ') + f'diff {d:.3e} scale {sc:.3g} {dict(key)}', flush=True)
    return r
ops.Ops.attention = both
x = synth.synth_clip(T=T, B=B, seed=1234).cuda()
net(x)
torch.cuda.synchronize()
print('done')
