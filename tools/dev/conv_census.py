"""Per-(kernel, shape) census of the keep_conv2d launches of one step at B clips (dev): time, TF/s, algorithmic GB/s.
conv_census.py [B] [family substring]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fam = sys.argv[2] if len(sys.argv) > 2 else 'conv_x3_kernel'
torch.cuda.set_device(0)
net, _ = bench.build_net(0, 1)
net.set_precision('x3')
x = bench.synth.synth_clip(T=20, B=B, seed=1234).cuda()
for _ in range(2):
    net(x)
torch.cuda.synchronize()
net.o.profile = []
net(x)
torch.cuda.synchronize()
rec, net.o.profile = net.o.profile, None
by = {}
for cfg, flops, split_k, e0, e1, nbytes, shape in rec:
    if fam not in cfg:
        continue
    d = by.setdefault((cfg, shape, split_k), [0.0, 0.0, 0, 0.0])
    d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += nbytes
tot = sum(v[1] for v in by.values())
print(f'family {fam}: {tot * 1e3:.1f} ms in {sum(v[2] for v in by.values())} launches')
print('kernel | (N,H,W,Cin,Cout,k,stride,up,pro) split | launches | ms total | us each | TF/s | GB/s algorithmic')
for (cfg, shape, sk), v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f'{cfg[:44]:44s} {str(shape):44s} sk{sk} n={v[2]:4d} {v[1] * 1e3:8.2f} ms {v[1] / v[2] * 1e6:8.1f} us {v[0] / v[1] / 1e12:7.1f} TF {v[3] / v[1] / 1e9:7.0f} GB/s')
