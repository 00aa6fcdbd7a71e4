"""dev: bench.py's end-to-end legs alone -- pipeline_leg (bench-side orchestration) next to product_leg (the product's
process_frames_u8 / process_image_sequence).   python tools/dev/product_leg.py [config3|config4|both] [frames]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'both'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
net = bench.KeepNet(**bench.DEFAULT_ARCH)
net.load_state_dict(bench.synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
cfgs = {'config3': (720, 1280, 1), 'config4': (1080, 1920, 3)}
for name, (H, W, f) in cfgs.items():
    if which not in (name, 'both'):
        continue
    print(name, 'pipeline_leg', json.dumps(bench.pipeline_leg(net, n, H, W, f)), flush=True)
    print(name, 'product_leg', json.dumps(bench.product_leg(net, n, H, W, f)), flush=True)
