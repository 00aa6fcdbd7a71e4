"""keep_absmax on the shapes the net probes (B = 16): contiguous tensors and channel slices of wider rows."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import ops

def bench(name, x, N, R, C, ld, img_stride):
    for _ in range(3):
        ops.absmax(x, N, R, C, ld, img_stride)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.absmax(x, N, R, C, ld, img_stride)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'{name:44s} {N * R * C * 4 / 1e6:8.1f} MB  {us:8.1f} us  {N * R * C * 4 / us / 1e6:6.2f} TB/s', flush=True)

N = 16
x = torch.randn(N, 256 * 256, 128, device='cuda')
bench('contiguous 16 x 256^2 x 128', x, N, 256 * 256, 128, 128, 256 * 256 * 128)
x = torch.randn(N, 256, 512, device='cuda')
bench('contiguous 16 x 256 tok x 512', x, N, 256, 512, 512, 256 * 512)
kv = torch.randn(N, 256, 1024, device='cuda')
bench('slice 512 of 1024-wide rows, 256 tok', kv, N, 256, 512, 1024, 256 * 1024)
x = torch.randn(N, 64 * 64, 256, device='cuda')
bench('contiguous 16 x 64^2 x 256', x, N, 64 * 64, 256, 256, 64 * 64 * 256)
x = torch.randn(320, 64 * 64, 128, device='cuda')
bench('contiguous 320 x 64^2 x 128', x, 320, 64 * 64, 128, 128, 64 * 64 * 128)
x = torch.randn(N, 512 * 512, 64, device='cuda')
bench('contiguous 16 x 512^2 x 64', x, N, 512 * 512, 64, 64, 512 * 512 * 64)
