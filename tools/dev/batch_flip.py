"""Batched vs solo forward of the config-3 test clips: first frame whose code indices differ for clip 0, and the logit margins
of the flipped tokens (chaos check).   python tools/dev/batch_flip.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
os.environ['KEEP_AMD_GRAPH'] = '0'
net = KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision(sys.argv[1] if len(sys.argv) > 1 else 'x3')
base = synth.ramp_image()
g = np.random.default_rng(300)
crops = [np.ascontiguousarray((np.roll(base, (7 * k) % 512, axis=1).astype(np.int16) + g.integers(-8, 9, (512, 512, 3))).clip(0, 255).astype(np.uint8))
         for k in range(160)]
u8 = torch.from_numpy(np.stack(crops)).cuda().view(8, 20, 512, 512, 3)
x = ((u8.float() / 255.) - 0.5) / 0.5
x = x.flip(-1).permute(0, 1, 4, 2, 3).contiguous()
o1, a1 = net(x[:1].contiguous(), need_upscale=False, return_aux=True)
o8, a8 = net(x, need_upscale=False, return_aux=True)
i1, i8 = a1['indices'][0].cpu(), a8['indices'][0].cpu()
m1 = a1.get('margins')
for t in range(20):
    d = (i1[t] != i8[t])
    pix = float((o1[0, t] - o8[0, t]).abs().max())
    msg = ''
    if d.any() and m1 is not None:
        msg = f' margins of flipped tokens (solo run): {m1[0][t][d].cpu().tolist()[:6]}'
    print(f'frame {t}: flipped tokens {int(d.sum())}, max |pixel diff| {pix:.3e}{msg}')
    if d.any():
        break
