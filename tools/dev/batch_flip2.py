import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
net = KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision('x3')
base = synth.ramp_image()
g = np.random.default_rng(300)
crops = [np.ascontiguousarray((np.roll(base, (7 * k) % 512, axis=1).astype(np.int16) + g.integers(-8, 9, (512, 512, 3))).clip(0, 255).astype(np.uint8))
         for k in range(300)]
clips = [torch.from_numpy(np.stack(crops[s:s + 20])) for s in range(0, 300, 20)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 15
print('clips_per_call', net.clips_per_call(20), 'using', nb)
solo = net.run_clips_u8(clips[:1], max_b=1)[0].numpy()
bat = net.run_clips_u8(clips[:nb], max_b=nb)[0].numpy()
bat2 = net.run_clips_u8(clips[:nb], max_b=nb)[0].numpy()
solo2 = net.run_clips_u8(clips[:1], max_b=1)[0].numpy()
print('batched run twice identical:', np.array_equal(bat, bat2), ' solo twice identical:', np.array_equal(solo, solo2))
for t in range(20):
    d = np.abs(solo[t].astype(np.int16) - bat[t].astype(np.int16))
    print(f'frame {t}: max diff {int(d.max())}, differing pixels {float((d > 0).mean()):.4f}, x3_fallbacks {net.x3_fallbacks}')
