import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import retinaface as RF
eng = RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0)).to('cuda')
g = torch.Generator().manual_seed(9)
frames = torch.randint(0, 256, (3, 320, 448, 3), generator=g, dtype=torch.uint8)
for thr in (0.9, 0.6):
    a = eng.detect_batch(frames, thr)
    eng.device_nms = False
    b = eng.detect_batch(frames, thr)
    eng.device_nms = True
    for i in range(3):
        same = a[i].shape == b[i].shape and np.array_equal(a[i], b[i])
        print(thr, i, a[i].shape, b[i].shape, 'equal' if same else 'DIFFER')
        if not same:
            n = min(len(a[i]), len(b[i]))
            bad = [k for k in range(n) if not np.array_equal(a[i][k], b[i][k])]
            print('  first differing rows', bad[:6])
            for k in bad[:4]:
                print('   dev ', a[i][k][:5]); print('   host', b[i][k][:5])
            sa = set(map(tuple, a[i].round(3)[:, :5].tolist())); sb = set(map(tuple, b[i].round(3)[:, :5].tolist()))
            print('  only dev', len(sa - sb), 'only host', len(sb - sa), 'tied scores in host', len(b[i]) - len(set(b[i][:, 4].tolist())))
