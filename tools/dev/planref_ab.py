"""dev: the plans' reference batch (Ops.plan_ref_images) against throughput at 16 clips per call and at ONE clip in flight."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import net as nm, synth  # noqa: E402
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH  # noqa: E402

net = nm.KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
x16 = synth.synth_clip(T=20, B=16, seed=1234).cuda()
x1 = x16[:1].contiguous()
for ref in [int(a) for a in sys.argv[1:]] or [16, 8, 4, 2, 16]:
    net.o.plan_ref_images = ref
    res = []
    for x, reps in ((x16, 3), (x1, 6)):
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res.append((x.shape[0], dt))
    print(f'plan_ref_images={ref:2d}: ' + '   '.join(f'B={b}: {dt * 1e3:8.2f} ms = {b * 20 / dt:6.1f} frames/s' for b, dt in res), flush=True)
