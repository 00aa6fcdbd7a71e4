import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L, ops
x = torch.randn(1, 16, 16, 64, device='cuda'); x[0, 3, 3, 5] = float('nan')
w = torch.randn(64, 1, 1, 64, device='cuda') * 0.1
for act in (0, L.ACT_RELU):
    y = ops.conv(x, w, None, pad=0, ksize=1, act=act)
    print('act', act, 'y[3,3,:4]', y[0, 3, 3, :4].tolist(), 'y[4,4,:2]', y[0, 4, 4, :2].tolist(), 'nan count', int(torch.isnan(y).sum()))
