"""RetinaFace on the engine at 640 x 1138, batch 16: network-only vs host-inclusive time (dev).  det_prof.py [x3|fp32] [resnet50|mobile0.25]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import retinaface as RF
prec = sys.argv[1] if len(sys.argv) > 1 else 'x3'
bb = sys.argv[2] if len(sys.argv) > 2 else 'resnet50'
det = RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0, backbone=bb), precision=prec).to('cuda')
frames = torch.randint(0, 256, (16, 640, 1138, 3), dtype=torch.uint8)
x = torch.rand((16, 640, 1138, 3), device='cuda') * 255 - 110
for _ in range(2):
    det.raw_outputs(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    det.raw_outputs(x)
torch.cuda.synchronize()
dn = (time.perf_counter() - t0) / 3
det.detect_batch(frames, 0.97)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    r = det.detect_batch(frames, 0.97)
torch.cuda.synchronize()
dd = (time.perf_counter() - t0) / 3
print(f'[{prec} {bb}] network only: {dn * 1e3:.1f} ms per 16 frames = {16 / dn:.1f} frames/s | detect_batch (host in/out): {dd * 1e3:.1f} ms = {16 / dd:.1f} frames/s | detections per frame {[len(a) for a in r][:4]}')
# section timing of one detect_batch call (synchronising between sections)
import numpy as np
def sect():
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    chunk = frames.to('cuda'); torch.cuda.synchronize(); t['H2D'] = time.perf_counter() - t0; t0 = time.perf_counter()
    xx = torch.empty((16, 640, 1138, 3), dtype=torch.float32, device='cuda')
    RF.L.call('keep_u8_to_f32', chunk, xx, chunk.numel()); xx = RF.ops.add_bcast(xx, det._mean, alpha=-1.0)
    heads = det.raw_heads(xx); torch.cuda.synchronize(); t['network'] = time.perf_counter() - t0; t0 = time.perf_counter()
    pri = torch.from_numpy(RF.prior_boxes(640, 1138)).cuda()
    dets = torch.empty((16, 4096, 16), device='cuda'); counts = torch.zeros(16, dtype=torch.int32, device='cuda')
    RF.L.call('keep_retina_decode', heads, pri, dets, counts, 16, heads.shape[1], 4096, 0.1, 0.2, 1138.0, 640.0, 0.97)
    cnt = counts.cpu().numpy(); rows = dets[:, :int(cnt.max())].cpu().numpy(); t['decode+D2H'] = time.perf_counter() - t0; t0 = time.perf_counter()
    for i in range(16):
        r = rows[i, :cnt[i]]; r = r[np.argsort(r[:, 15], kind='stable')]; r = r[r[:, 4].argsort()[::-1]]
        RF.nms(np.ascontiguousarray(r[:, :5]), 0.4)
    t['host sort+nms'] = time.perf_counter() - t0
    return t
sect(); print({k: round(v * 1e3, 2) for k, v in sect().items()})
