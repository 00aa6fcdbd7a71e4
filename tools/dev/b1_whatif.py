"""dev: what does a class of small launches cost ONE clip in flight?  Every listed entry point is issued TWICE (they are idempotent: same inputs,
same outputs -- the data, and with it the clock the chip grants the matrix kernels, stay what they are), the forward is re-captured and timed by
hipGraph replay: the difference to the plain forward is what those launches cost in situ, i.e. the UPPER BOUND of what folding them into their
producers / consumers could return (VERDICT r5 item 1c pricing).  (`skip` mode -- the launches turned into no-ops -- is kept for reference: it
poisons the data with NaNs, the matrix kernels then draw less power and the whole clip runs 10 % faster: NOT a price.)
   python tools/dev/b1_whatif.py [skip]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L, net as net_mod, synth  # noqa: E402
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH  # noqa: E402

net = net_mod.KeepNet(**DEFAULT_ARCH)
net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
x = synth.synth_clip(T=20, B=1, seed=1234).cuda()
real_call = L.call
MODE = sys.argv[1] if len(sys.argv) > 1 else 'twice'
skipped = {}


def timed(label, skip):
    def call(name, *args):
        if name in skip:
            skipped[name] = skipped.get(name, 0) + 1
            if MODE == 'skip':
                return
            real_call(name, *args)
        real_call(name, *args)
    L.call = call
    net_mod.L.call = call
    import comfyui_keep_amd.engine.ops as ops_mod
    ops_mod.L.call = call
    net._graphs.clear(); net._graph_seen.clear(); skipped.clear()
    net(x, _defer_check=True)                      # eager pass: every launch goes through `call` once
    per_pass = dict(skipped)
    for _ in range(2):                             # capture, first replay
        net(x, _defer_check=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        net(x, _defer_check=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    word = 'skipped' if MODE == 'skip' else 'doubled'
    print(f'{label:60s} {dt * 1e3:8.2f} ms per clip  {20 / dt:6.1f} frames/s   {word} per pass: {per_pass}', flush=True)
    return dt


base = timed('full forward', set())
for label, skip in (('norm_finalize x 2', {'keep_norm_finalize'}),
                    ('small GroupNorm statistics x 2', {'keep_group_stats'}),
                    ('LayerNorm (all forms) x 2', {'keep_layernorm', 'keep_layernorm_amax'}),
                    ('range probes x 2', {'keep_absmax'}),
                    ('geglu / concat2 / add_bcast / kalman_update / flow_warp x 2', {'keep_geglu', 'keep_geglu_amax', 'keep_concat2', 'keep_add_bcast', 'keep_kalman_update', 'keep_flow_warp'}),
                    ('ALL of the above x 2', {'keep_norm_finalize', 'keep_group_stats', 'keep_layernorm', 'keep_layernorm_amax', 'keep_absmax', 'keep_geglu',
                                               'keep_geglu_amax', 'keep_concat2', 'keep_add_bcast', 'keep_kalman_update', 'keep_flow_warp'}),
                    ('full forward again', set())):
    timed(label, skip)
