cd /root/repo
timeout 900 python -m pytest tests/test_gpu_facelib.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | tail -30
import sys, time, types
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L
from comfyui_keep_amd.engine import retinaface as RF
def each(f, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) * 1e3, 1))
    return out
fr = torch.randint(0, 256, (16, 640, 1138, 3), dtype=torch.uint8)
for bb in ('mobile0.25', 'resnet50'):
    rf = RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0, backbone=bb)).to('cuda')
    r = rf.detect_batch(fr, 0.97)
    print(bb, 'faces per frame', [len(q) for q in r][:8])
    print(bb, 'detect_batch', each(lambda: rf.detect_batch(fr, 0.97)))
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable(); rf.detect_batch(fr, 0.97); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(8); print('\n'.join(s.getvalue().splitlines()[6:18]))
PY
