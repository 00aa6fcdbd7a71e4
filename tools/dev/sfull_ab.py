#!/usr/bin/env python
"""A/B of the 32-query full-score-row attention kernel (AttnBlock at 16x16: 256 tokens, one head of d = 512) against the
128-query one (keep_attention_args.flags = KEEP_ATTN_NO_SFULL2): error vs an fp64 softmax(QK^T)V and time, B = 1 and 16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L  # noqa: E402
from comfyui_keep_amd.engine import ops  # noqa: E402

for B in (1, 16):
    torch.manual_seed(B)
    Ltok, C = 256, 512
    qkv = torch.randn(B * Ltok, 3 * C, device='cuda')
    s3 = (Ltok * 3 * C, 3 * C, 0)
    q, k, v = (qkv.view(B, Ltok, 3, C)[:, :, i].double() for i in range(3))
    ref = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, -1) @ v
    for mode in ('old', 'new'):
        if mode == 'old':
            ops.DEFAULT.attn_flags = L.ATTN_NO_SFULL2
        else:
            ops.DEFAULT.attn_flags = 0
        o = torch.empty(B * Ltok, C, device='cuda')

        def run():
            ops.attention(qkv, ops.offset(qkv, C), ops.offset(qkv, 2 * C), o, B=B, H=1, Lq=Ltok, Lk=Ltok, D=C, Dv=C, scale=C ** -0.5,
                          q_str=s3, k_str=s3, v_str=s3, o_str=(Ltok * C, C, 0), mma=L.MMA_X3)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        err = (o.view(B, Ltok, C).double() - ref).abs().max().item()
        print(f'B={B:2d} {mode}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us   max err vs fp64 {err:.2e}', flush=True)
