"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 (B=1, T=2, 512x512, synthetic
weights) checked against the CPU oracle run on the same inputs (and the reference-generated golden frame 0)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


def run_smoke():
    import keep_oracle as O
    from comfyui_keep_amd.engine import synth
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
    from comfyui_keep_amd.engine.net import KeepNet
    assert torch.cuda.is_available(), "smoke() needs cuda:0 (MI355X)"
    W = synth.synth_state_dict(seed=0)
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(W, strict=True)
    net.to('cuda:0').eval()
    x = synth.synth_clip(T=2, B=1, seed=1234)
    t0 = time.time()
    out, aux = net(x.cuda(), need_upscale=False, return_aux=True)
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    ref, raux = O.keep_forward(x, W, return_aux=True)
    top2 = raux['logits'].topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    agree = aux['indices'].cpu().long() == raux['indices']
    assert agree[safe].all(), f"code indices differ on {int((~agree[safe]).sum())} safe tokens"
    forced = net(x.cuda(), force_indices=raux['indices'].to(torch.int32))
    err = (forced.cpu() - ref).abs().max().item()
    print(f"smoke: T=2 512x512 HIP forward {t_gpu:.2f}s (first call), index agreement {agree.float().mean():.4f}, "
          f"max-abs diff vs oracle (oracle indices) {err:.2e}")
    assert err <= 1e-3, err
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'keep_forward_T3.npz'))
    assert np.array_equal(aux['indices'][0, 0].cpu().numpy().astype(np.int16)[g['margins'][0] > 1e-3],
                          g['indices'][0][g['margins'][0] > 1e-3])
    return err
