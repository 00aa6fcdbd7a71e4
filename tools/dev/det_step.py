"""dev: the detector networks alone, a few calls at 16 frames (for rocprofv3 --kernel-trace --stats).  det_step.py resnet50|mobile0.25|YOLOv5n|YOLOv5l"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import retinaface as RF, yoloface as YF  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'resnet50'
if which.startswith('YOLO'):
    eng = YF.YoloFaceEngine(YF.synth_yolo_state_dict(which, seed=0)).to('cuda')
    x = torch.rand((16, 704, 1152, 3), device='cuda')
    run = lambda: eng.forward_nhwc(x)
else:
    eng = RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0, backbone=which)).to('cuda')
    x = torch.rand((16, 640, 1138, 3), device='cuda') * 255 - 110
    run = lambda: eng.raw_heads(x)
for _ in range(4):
    run()
torch.cuda.synchronize()
