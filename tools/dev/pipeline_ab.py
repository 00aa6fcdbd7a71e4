#!/usr/bin/env python
"""bench.py's end-to-end leg (frames -> detect -> crop -> restore -> parse -> paste -> frames) on its own, repeated (dev):
python tools/dev/pipeline_ab.py [frames] [H] [W] [faces] [repeats]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
a = [int(v) for v in sys.argv[1:]] + [300, 1080, 1920, 3, 3][len(sys.argv) - 1:]
torch.cuda.set_device(0)
net, _ = bench.build_net(0, 1)
net.set_precision('x3')
for r in range(a[4]):
    res = bench.pipeline_leg(net, a[0], a[1], a[2], a[3])
    print('PIPE', r, res['value'], res['seconds'], flush=True)
