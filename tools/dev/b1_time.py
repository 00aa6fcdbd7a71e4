"""dev: ms per clip with ONE clip in flight (hipGraph replay), two-stream order on / off, chunk sizes.
   python tools/dev/b1_time.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import net as net_mod, synth  # noqa: E402
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = net_mod.KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
ref = None
for name, mx, first, chunk in (('one stream', 0, 3, 4), ('two streams 3/4', 2, 3, 4), ('two streams 2/3', 2, 2, 3), ('two streams 4/5', 2, 4, 5),
                               ('two streams 1/2', 2, 1, 2), ('two streams 3/8', 2, 3, 8), ('one stream', 0, 3, 4)):
    net_mod.STREAM_OVERLAP_MAX_CLIPS, net_mod.STREAM_OVERLAP_FIRST, net_mod.STREAM_OVERLAP_CHUNK = mx, first, chunk
    net._graphs.clear(); net._graph_seen.clear()
    for _ in range(3):
        out = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    if ref is None:
        ref = out.clone()
    print(f'B={B} {name:18s} {dt * 1e3:8.2f} ms per call  {B * 20 / dt:7.1f} frames/s  bit-identical to the first: {torch.equal(out, ref)}  graphs {len(net._graphs)}', flush=True)
