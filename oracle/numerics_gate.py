"""TEST INFRASTRUCTURE ONLY (CPU, no kernel): a numerics gate for the two levers that could lift the x3 convolutions above the
three-MFMAs-per-product ceiling.  Each variant re-routes some 3x3 convolutions of the CPU
oracle (oracle/keep_oracle.py, pinned to the reference) through an EMULATION of the cheaper arithmetic and runs the reference goldens of
tests/golden/ under the assertions tests/test_gpu_net.py::_full_forward_check applies to the HIP path:

  baseline        the oracle as it is (fp32 direct convolution) -- what x3 is equivalent to (2^-22 per product)
  winograd64      Winograd F(2x2, 3x3) (Lavin & Gray 2016) evaluated in fp32 for every stride-1 3x3 convolution with Cout = 64 and >= 64 input
                  channels (the 64 ch @512^2 layers: 36 MACs per 2x2 outputs become 16 -- 2.25 x fewer products): input tiles B^T d B, weights
                  G g G^T, channel sum of element-wise products in fp32 (the MFMA accumulator), outputs A^T m A
  winograd_hires  the same for every stride-1 3x3 convolution on maps of >= 256^2 pixels (adds the 128 ch @256^2 layers)
  mxfp8_lo_18_22  generator blocks 18-22 (KA / VQ:339-343: everything behind the last Upsample but one; 25 % of all FLOPs; nothing discrete
                  follows inside the frame): the x3 product a.w = a_hi.w_hi + a_hi.w_lo + a_lo.w_hi with the two LOW terms on MX-fp8 operands
                  (OCP MX: 32-element blocks along the reduction axis sharing one power-of-two scale, e4m3 elements) -- on gfx950 those two
                  MFMAs would run at twice the fp16 rate (2 instead of 3 units per product)
  mxfp8_lo_all    the same for EVERY convolution of the two encoders, the generator and the CFT blocks (the index chain included)

    python oracle/numerics_gate.py [variant ...] [--t20]      # T = 3 always; --t20 adds the metric's own clip length (~2 min per variant)

Prints one JSON row per (variant, golden): index agreement / first frame with a flip, top-1 logit error up to it, max-abs pixel difference
of the frames in front of it (free running), and the per-frame pixel difference with the reference's indices injected -- and whether the
row passes the product's assertions unchanged (``go``).
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import keep_oracle as O  # noqa: E402
from comfyui_keep_amd.engine import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
LOGIT_BOUND = {'keep_forward_T3.npz': 2.8e-3, 'keep_forward_T20.npz': 6.6e-3}       # tests/test_gpu_net.py:LOGIT_ERR_BOUNDS

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3(x, w, bias):
    """F(2x2, 3x3), stride 1, zero padding 1, all arithmetic fp32: x [N,C,H,W] (H, W even), w [O,C,3,3]."""
    N, C, H, W = x.shape
    if N > 1:                                                                # one image at a time: the tile tensors of a 512^2 map are 1-2 GB each
        return torch.cat([winograd_conv3x3(x[n:n + 1], w, bias) for n in range(N)], 0)
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [N,C,H/2,W/2,4,4]
    V = torch.einsum('ij,nchwjk,lk->nchwil', BT, d, BT)                      # B^T d B
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)                            # G g G^T  [O,C,4,4]
    M = torch.einsum('ocil,nchwil->nohwil', U, V)                            # channel sum of element-wise products (fp32 accumulate)
    Y = torch.einsum('ij,nohwjk,lk->nohwil', AT, M, AT)                      # A^T m A  [N,O,H/2,W/2,2,2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def _pow2_floor(t):
    return torch.exp2(torch.floor(torch.log2(t.clamp_min(1e-30))))


def mx_fp8(t, axis):
    """OCP MX-fp8 (e4m3) quantise -> dequantise along ``axis`` in blocks of 32: shared scale 2^(floor(log2 amax) - 8), elements rounded to
    float8_e4m3fn (max normal 448 = 1.75 * 2^8)."""
    t = t.movedim(axis, -1)
    shp = t.shape
    pad = (-shp[-1]) % 32
    tt = F.pad(t, (0, pad)).reshape(*shp[:-1], -1, 32)
    amax = tt.abs().amax(-1, keepdim=True)
    scale = _pow2_floor(amax) * 2.0 ** -8
    scale = torch.where(amax > 0, scale, torch.ones_like(scale))
    q = (tt / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * scale      # (the OCP conversion saturates; torch's cast gives NaN)
    return q.reshape(*shp[:-1], -1)[..., :shp[-1]].movedim(-1, axis)


def split16(t, scale):
    hi = (t * scale).to(torch.float16).to(torch.float32)
    lo = (t * scale - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def x3_conv_mxfp8_low_terms(x, w, bias, stride, padding):
    """The x3 product with its two low terms on MX-fp8 operands.  Range scales as the product's: activations per image into [2^14, 2^15),
    weights per tensor just below 2^15 (engine/ops.py:x3_scale_for), de-scaled on the result."""
    sa = 2.0 ** (14 - torch.ceil(torch.log2(x.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30))))
    sw = 2.0 ** (14 - torch.ceil(torch.log2(w.abs().max().clamp_min(1e-30))))
    a_hi, a_lo = split16(x, sa)
    w_hi, w_lo = split16(w, sw)
    y = F.conv2d(a_hi, w_hi, None, stride=stride, padding=padding)
    y = y + F.conv2d(mx_fp8(a_hi, 1), mx_fp8(w_lo, 1), None, stride=stride, padding=padding)
    y = y + F.conv2d(mx_fp8(a_lo, 1), mx_fp8(w_hi, 1), None, stride=stride, padding=padding)
    y = y / (sa * sw)
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def make_conv(variant, counter):
    base = O.conv.__wrapped__ if hasattr(O.conv, '__wrapped__') else O.conv

    def conv(x, W, p, stride=1, padding=1):
        w, b = W[f'{p}.weight'], W.get(f'{p}.bias')
        k3s1 = w.shape[-1] == 3 and stride == 1 and padding == 1 and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0
        if variant == 'winograd64' and k3s1 and w.shape[0] == 64 and w.shape[1] >= 64:
            counter[0] += 1
            return winograd_conv3x3(x, w, b)
        if variant == 'winograd_hires' and k3s1 and w.shape[1] >= 64 and x.shape[-1] * x.shape[-2] >= 256 * 256:
            counter[0] += 1
            return winograd_conv3x3(x, w, b)
        if variant == 'mxfp8_lo_18_22' and w.shape[-1] == 3 and w.shape[1] % 32 == 0 and any(p.startswith(f'generator.blocks.{j}.') for j in (18, 19, 20, 21, 22)):
            counter[0] += 1
            return x3_conv_mxfp8_low_terms(x, w, b, stride, padding)
        if variant == 'mxfp8_lo_all' and w.shape[1] % 32 == 0:        # every VQGAN-stack / CFT convolution (3x3 and 1x1) of both encoders and the generator
            counter[0] += 1
            return x3_conv_mxfp8_low_terms(x, w, b, stride, padding)
        return base(x, W, p, stride, padding)
    conv.__wrapped__ = base
    return conv


def digest(frames):
    H, Wd = frames.shape[-2:]
    return frames[:, :, 7::H // 32, 5::Wd // 32][:, :, :32, :32]


def run(variant, gold_name, T, W):
    g = np.load(os.path.join(GOLD, gold_name))
    x = synth.synth_clip(T=T, B=1, seed=1234, pattern='texture')
    counter = [0]
    keep = O.conv
    O.conv = make_conv(variant, counter)
    try:
        t0 = time.time()
        out, aux = O.keep_forward(x, W, return_aux=True)
        forced = torch.from_numpy(g['indices'].astype(np.int64)).view(1, T, -1)
        out_f = O.keep_forward(x, W, force_indices=forced)
        secs = time.time() - t0
    finally:
        O.conv = keep
    idx = aux['indices'][0].numpy().astype(np.int16)
    agree = idx == g['indices']
    first = next((t for t in range(T) if not agree[t].all()), T)
    upto = min(first + 1, T)
    top1 = aux['logits'][0].max(-1).values.numpy()
    dl = np.abs(top1[:upto] - g['logit_top1'][:upto])
    dl_agree = float(dl[agree[:upto]].max())
    free = np.abs(digest(out[0]).numpy() - g['out_grid']).reshape(T, -1).max(1)
    inj = np.abs(digest(out_f[0]).numpy() - g['out_grid']).reshape(T, -1).max(1)
    scale = float(np.abs(g['out_grid']).max())
    flips = g['margins'][first][~agree[first]].tolist() if first < T else []
    ok = (agree[0][g['margins'][0] > 1e-3].all() and float(dl[0].max()) <= 1e-3 and dl_agree <= LOGIT_BOUND[gold_name]
          and all(agree[t][g['margins'][t] > max(1e-3, 2 * dl_agree)].all() for t in range(upto)) and first >= 1
          and all(m <= 2 * dl_agree for m in flips) and (first == 0 or float(free[:first].max()) <= min(1e-3, 3e-4 * scale))
          and float(free[0]) <= 5e-5 * scale and float(inj.max()) <= 1e-3 and float(inj.max()) <= 3e-4 * scale)
    row = {'variant': variant, 'golden': gold_name, 'convolutions_rerouted_per_run': counter[0] // 2, 'seconds': round(secs, 1),
           'index_agreement': round(float(agree.mean()), 5), 'first_frame_with_a_flip': first,
           'frame0_top1_logit_err': float(dl[0].max()), 'max_top1_logit_err_up_to_first_flip': dl_agree,
           'margins_of_first_flips': [round(float(m), 6) for m in flips],
           'free_running_pixel_err_before_first_flip': [round(float(v), 7) for v in free[:first]],
           'pixel_err_reference_indices_injected_max': float(inj.max()), 'pixel_err_injected_frame0': float(inj[0]),
           'output_scale': round(scale, 3), 'go': bool(ok)}
    print(json.dumps(row), flush=True)
    return row


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    variants = args or ['baseline', 'winograd64', 'winograd_hires', 'mxfp8_lo_18_22']
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    W = synth.synth_state_dict(seed=0)
    for v in variants:
        run(v, 'keep_forward_T3.npz', 3, W)
        if '--t20' in sys.argv:
            run(v, 'keep_forward_T20.npz', 20, W)


if __name__ == '__main__':
    main()
