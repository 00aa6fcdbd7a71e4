import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L, ops
torch.manual_seed(0)
x = torch.randn(1, 48, 48, 128, device='cuda'); w = torch.randn(192, 1, 1, 128, device='cuda') * 0.1; b = torch.randn(192, device='cuda')
sc = ops.x3_scale_for(float(w.abs().max())); wx3 = ops.split_x3(w.reshape(-1, 128), sc).view(-1)
for env in (None, '1024'):
    if env: os.environ['KEEP_GATHER_SMALL_M'] = env
    for mma in (L.MMA_F32, L.MMA_X3):
        kw = dict(pad=0, ksize=1, stats=True, mma=mma, split_k=1)
        if mma == L.MMA_X3: kw.update(wx3=wx3, x3_acc_scale=1.0 / sc)
        ops.DEFAULT.profile = []
        y, st = ops.conv(x, w, b, **kw)
        name = ops.DEFAULT.profile[-1][0]; ops.DEFAULT.profile = None
        torch.cuda.synchronize()
        if st is None:
            print(env, mma, name, 'no stats'); continue
        print(env, mma, name, 'P', st.P, 'part', None if st.part is None else tuple(st.part.shape), 'amax', st.amax)
        if st.part is not None:
            rows = 2304 // st.P
            yy = y.reshape(st.P, rows, 192)
            ref = torch.stack([yy.sum(1), (yy * yy).sum(1)], -1)
            d = (st.part[0] - ref).abs()
            print('   partial err max', float(d.max()), 'bad partial rows', sorted(set(torch.nonzero(d.amax((1, 2)) > 1e-2).flatten().tolist()))[:20],
                  'bad couts', sorted(set(torch.nonzero(d.amax((0, 2)) > 1e-2).flatten().tolist()))[:20])
