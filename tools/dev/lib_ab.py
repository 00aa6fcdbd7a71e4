#!/usr/bin/env python
"""A/B of whole-step time between builds of libkeep_hip.so (dev): one subprocess per library (KEEP_HIP_LIB), rounds alternate.
python tools/dev/lib_ab.py [--b 16] [--rounds 2] name=path.so ...      (path 'default' = the in-tree library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(B, passes):
    import torch
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from comfyui_keep_amd.engine import synth
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
    from comfyui_keep_amd.engine.net import KeepNet
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
    net.to('cuda').eval().set_precision('x3')
    x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
    for _ in range(2):
        out = net(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(passes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = net(x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('AB', ' '.join(f'{t:.1f}' for t in ts), 'digest', float(out.double().abs().sum()))


if __name__ == '__main__':
    a = sys.argv[1:]
    if a and a[0] == '--child':
        child(int(a[1]), int(a[2]))
        sys.exit(0)
    B, rounds, libs = 16, 2, []
    i = 0
    while i < len(a):
        if a[i] == '--b':
            B = int(a[i + 1]); i += 2
        elif a[i] == '--rounds':
            rounds = int(a[i + 1]); i += 2
        else:
            n, p = a[i].split('='); libs.append((n, p)); i += 1
    for r in range(rounds):
        for n, p in libs:
            env = dict(os.environ)
            if p != 'default':
                env['KEEP_HIP_LIB'] = os.path.abspath(p)
            res = subprocess.run([sys.executable, __file__, '--child', str(B), '3'], env=env, capture_output=True, text=True)
            line = [l for l in res.stdout.splitlines() if l.startswith('AB')]
            print(f'round {r} {n:12s}', line[0] if line else ('FAILED ' + res.stderr[-400:]), flush=True)
