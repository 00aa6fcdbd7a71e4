import os, sys, torch
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
net = KeepNet(**DEFAULT_ARCH); net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
for prec in ('x3', 'fp32'):
    net.set_precision(prec)
    for T in (1, 2):
        x = synth.synth_clip(T=T, B=1, seed=5).cuda()
        net.graph_mode = '0'
        a = net(x); b = net(x)
        net.graph_mode = '1'
        c = net(x); d = net(x)
        print(prec, 'T', T, 'eager-eager', float((a - b).abs().max()), 'eager-graph', float((a - c).abs().max()), 'graph-graph', float((c - d).abs().max()))
