"""dev: cProfile of the product's sequence entry point (process_frames_u8) on BASELINE configs[2]: where the host time of the streamed path goes.
   python tools/dev/product_prof.py [frames]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench  # noqa: E402
import torch  # noqa: E402
import synth_facehelper as SF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, W, faces = 720, 1280, 1
net = bench.KeepNet(**bench.DEFAULT_ARCH)
net.load_state_dict(bench.synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
proc, helper = SF.make_processor(net, (H, W), faces)
g = torch.Generator().manual_seed(0)
frames = [torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).numpy() for _ in range(n)]
for rep in range(2):
    helper.begin_sequence()
    t0 = time.perf_counter()
    proc.process_frames_u8(frames, 1.0, False, True, False, max_clip_length=20)
    torch.cuda.synchronize()
    print('warm pass', rep, round(time.perf_counter() - t0, 3), 's', flush=True)
helper.begin_sequence()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
proc.process_frames_u8(frames, 1.0, False, True, False, max_clip_length=20)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print('profiled pass', round(dt, 3), 's =', round(n / dt, 1), 'video frames/s')
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
