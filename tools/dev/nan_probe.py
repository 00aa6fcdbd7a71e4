"""Find kernels that read memory nobody wrote: every fresh float buffer is NaN-filled, KEEP_DEBUG_SYNC names the first op
whose output turns non-finite."""
import os, sys
os.environ['KEEP_DEBUG_SYNC'] = '1'
import torch
_empty, _empty_like = torch.empty, torch.empty_like
def empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float('nan'))
    elif t.is_cuda and t.dtype in (torch.int32, torch.int16, torch.uint8): t.fill_(113)
    return t
def empty_like(x, **k):
    t = _empty_like(x, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float('nan'))
    return t
torch.empty, torch.empty_like = empty, empty_like
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
net = KeepNet(**DEFAULT_ARCH); net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision(sys.argv[1] if len(sys.argv) > 1 else 'x3')
net.graph_mode = '0'
x = synth.synth_clip(T=2, B=1, seed=5).cuda()
out = net(x)
print('finite:', bool(torch.isfinite(out).all()))
