"""GPU suite (-m gpu): module-level and whole-network parity of the HIP engine (engine/net.py, every op through
the C-ABI) against (a) the committed golden vectors generated from the imported reference and (b) the CPU oracle
on the same seeded inputs.

Tolerance (fp32-in / fp32-accumulate MFMA policy): max-abs <= 1e-3 on network outputs (BASELINE.json north_star),
tighter per module.  Code indices: bit-exact wherever the reference's top-1/top-2 logit margin exceeds 1e-3
(exact ties are undefined behaviour in the reference: topk documents no order).
"""
import os

import numpy as np
import pytest
import torch

import keep_oracle as O
from conftest import GOLDEN, op_input
from comfyui_keep_amd.engine import arch, ops, synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH, encoder_blocks, generator_blocks

pytestmark = pytest.mark.gpu
# Largest |top-1 logit - reference's| tolerated on the frames of a free-running clip up to (and at) its first index flip.  Frame 0
# sees no optical flow and is held to 1e-3.  Later frames see GMFlow's flows -- 4096-way soft-arg-maxes whose fp32 re-association
# moves a flow by ~1e-3 px -- through the warped previous output, and the synthetic net's pixel-level texture turns that into a
# logit shift of a few 1e-3, for the exact-f32 policy and the x3 policy alike.
#   physical regime (round 4: engine/synth.py, flows of median 1 px / p99 6-13 px): measured 1.4e-3 (T=3), 3.3e-3 / 3.0e-3 (T=20,
#     x3 / fp32), 6.2e-3 (Asian T=2, frames of +-3.8)                                      -> bound 8e-3
#   wide regime (round 3's weights + clip, flows of hundreds of pixels: the out-of-range edge case): 8.3e-3 ... 1.7e-2  -> 2.5e-2
# The flip rule is NOT this constant: a token may differ from the reference only if its margin is below twice the top-1 logit
# error MEASURED in the same run (`per_frame_top1_logit_err` is printed by every run).
# Round 6: the bound is PER GOLDEN, twice the largest value ever measured on it (any box, either policy) -- a single 8e-3 for the whole
# regime admitted flips up to margin 1.6e-2 on the goldens that never came near it.
LOGIT_ERR_BOUND = 8e-3            # (the physical regime's ceiling: no golden of it may exceed this)
LOGIT_ERR_BOUND_WIDE = 2.5e-2
LOGIT_ERR_BOUNDS = {'keep_forward_T3.npz': 2.8e-3,          # measured 1.4e-3
                    'keep_forward_T20.npz': 6.6e-3,         # measured 3.3e-3 / 3.0e-3 (x3 / fp32)
                    'keep_forward_asian_T2.npz': 8e-3,      # measured 6.2e-3 (frames of +-3.8: 1.3 x -- at the regime's ceiling)
                    'keep_forward_T3_wide.npz': 2.5e-2}     # measured 8.3e-3 ... 1.7e-2
OPS = np.load(os.path.join(GOLDEN, 'ops.npz'))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def close(got, ref, tol, what):
    ref = torch.from_numpy(ref) if isinstance(ref, np.ndarray) else ref
    got = got.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, f'{what}: max abs err {err:.3e} (scale {scale:.3g}, tol {tol})'


def test_resblock_down_up_attn_vs_golden(gpu_net):
    net = gpu_net
    close(nchw(net._resblock(nhwc(op_input('res_same', (1, 128, 16, 16))), 'encoder.blocks.5')[0]), OPS['res_same'], 3e-4, 'res')
    close(nchw(net._resblock(nhwc(op_input('res_proj', (1, 64, 16, 16))), 'encoder.blocks.4')[0]), OPS['res_proj'], 3e-4, 'res proj')
    w = net.w
    y = net.o.conv(nhwc(op_input('down', (1, 128, 16, 16))), w['encoder.blocks.6.conv.weight'], w['encoder.blocks.6.conv.bias'], down=True)
    close(nchw(y), OPS['down'], 3e-4, 'down')
    y = net.o.conv(nhwc(op_input('up', (1, 128, 8, 8))), w['generator.blocks.17.conv.weight'], w['generator.blocks.17.conv.bias'], upsample=True)
    close(nchw(y), OPS['up'], 3e-4, 'up')
    close(nchw(net._attnblock(nhwc(op_input('attn', (1, 512, 8, 8))), 'encoder.blocks.17')), OPS['attn'], 3e-4, 'attnblock')


def test_cft_cfa_vs_golden(gpu_net):
    net = gpu_net
    y, _ = net._cft(nhwc(op_input('cft_enc', (1, 256, 8, 8))), nhwc(op_input('cft_dec', (1, 256, 8, 8))), 'cft.32')
    close(nchw(y), OPS['cft'], 3e-4, 'cft')
    y = net._cfa(nhwc(op_input('cfa_curr', (1, 256, 8, 8))), nhwc(op_input('cfa_prev', (1, 256, 8, 8))), 'cfa.32')
    close(nchw(y), OPS['cfa'], 3e-4, 'cfa')


def test_cft_encoder_half_once_per_clip_equals_the_whole_block(gpu_net):
    """Round 6: encode_enc's first convolution and 1x1 shortcut over cat[enc, dec] (KA:466, VQ:170-181) as encoder half (once per clip, all
    frames: `_cft_enc_part`) + decoder half (frame loop): the GroupNorm over 2C channels never mixes the halves and both operators are sums
    over input channels.  Against the one-convolution form (KEEP_CFT_SPLIT=0) to fp32 re-association; the precomputed halves handed in
    equal the on-the-fly ones bit for bit; statistics from a producer (dec_st) equal the probed ones to rounding."""
    from comfyui_keep_amd.engine import net as net_mod
    net = gpu_net
    g = torch.Generator().manual_seed(11)
    for key, C, hw in (('32', 256, 32), ('16', 512, 16)):
        enc = (torch.randn((2, hw, hw, C), generator=g) * 1.5).cuda()
        dec = (torch.randn((2, hw, hw, C), generator=g) * 0.7 + 0.2).cuda()
        y, st = net._cft(enc, dec, f'cft.{key}')
        pre = net._cft_enc_part(enc, f'cft.{key}')
        y2, _ = net._cft(enc, dec, f'cft.{key}', pre)
        assert torch.equal(y, y2)
        old = net_mod.CFT_SPLIT
        try:
            net_mod.CFT_SPLIT = False
            ref, _ = net._cft(enc, dec, f'cft.{key}')
        finally:
            net_mod.CFT_SPLIT = old
        scale = float(ref.abs().max())
        err = float((y - ref).abs().max())
        print(f'cft.{key} [{net.precision}] split vs one convolution: {err:.3e} of scale {scale:.3g}')
        assert torch.isfinite(y).all() and err <= 3e-6 * max(1.0, scale)


def test_cfa_range_scales_from_producer_maxima(gpu_net):
    """x3 policy: CFA takes five of its nine range scales from maxima its producers already hold (`curr_amax`: the producing
    convolution's fused max|out|; the q / kv projections' fused maxima; |attention output| <= max|v|) instead of probing -- with inputs of
    the Asian config's residual-stream magnitude (1e4 .. 6e4) the result stays the probed path's to fp32 rounding, and a loose but valid
    bound (4 x) costs at most two bits of the low halves."""
    from comfyui_keep_amd.engine import net as net_mod, ops
    net = gpu_net
    if net.precision != 'x3':
        pytest.skip('range scales are an x3 matter')
    curr = nhwc(op_input('cfa_curr', (2, 256, 8, 8))) * torch.tensor([3.0e4, 1.0]).view(2, 1, 1, 1).cuda()
    prev = nhwc(op_input('cfa_prev', (2, 256, 8, 8))) * torch.tensor([1.0e4, 5.0]).view(2, 1, 1, 1).cuda()
    amax = curr.abs().flatten(1).max(1).values
    old = net_mod.CFA_FREE_RANGES
    try:
        net_mod.CFA_FREE_RANGES = False
        ref = net._cfa(curr, prev, 'cfa.32')
        net_mod.CFA_FREE_RANGES = True
        got = net._cfa(curr, prev, 'cfa.32', amax.contiguous())
        loose = net._cfa(curr, prev, 'cfa.32', (amax * 4).contiguous())
        # the previous frame's maximum handed in as well (what the frame loop does from frame 1 on): the kv projection's probe goes too --
        # an exact maximum, so the result is the same bits; the returned max|z| is the probe's value
        pmax = prev.abs().flatten(1).max(1).values.contiguous()
        got2, z_amax = net._cfa(curr, prev, 'cfa.32', amax.contiguous(), pmax, want_amax=True)
    finally:
        net_mod.CFA_FREE_RANGES = old
    assert torch.equal(got2, got) and torch.equal(z_amax, got.abs().flatten(1).max(1).values)
    assert torch.isfinite(got).all()
    sc = ref.abs().flatten(1).max(1).values.view(2, 1, 1, 1)
    assert float(((got - ref).abs() / sc).max()) <= 2e-6
    assert float(((loose - ref).abs() / sc).max()) <= 1e-5


def test_kalman_gain_vs_golden_and_batched(gpu_net, synth_weights):
    z = op_input('kalman_z', (1, 3, 256, 8, 8))
    g = gpu_net._kalman_gain(nhwc(z[0]), 1, 3)
    close(g.view(1, 3, 1, 8, 8), OPS['kalman_gain'], 2e-4, 'kalman gain')
    # two clips on the batch axis == each clip alone (clips share no state)
    z2 = torch.cat([z, op_input('kalman_z2', (1, 3, 256, 8, 8))], 0)
    g2 = gpu_net._kalman_gain(nhwc(z2.flatten(0, 1)), 2, 3).view(2, 3, 64)
    ref = O.kalman_calc_gain(z2, synth_weights, DEFAULT_ARCH).view(2, 3, 64)
    close(g2, ref, 2e-4, 'kalman gain B=2')


def test_code_prediction_vs_oracle(gpu_net, synth_weights):
    z = op_input('codes_z', (2, 256, 16, 16), 1.5)
    quant, idx, margin = gpu_net._predict_codes(nhwc(z), want_aux=True)
    logits, ref_idx = O.predict_codes(z, synth_weights, DEFAULT_ARCH)
    top2 = logits.topk(2, -1).values
    ref_margin = top2[..., 0] - top2[..., 1]
    safe = ref_margin > 1e-3
    assert safe.float().mean() > 0.9
    assert torch.equal(idx.cpu().long()[safe], ref_idx[safe])
    close(margin, ref_margin, 2e-4, 'logit margins')
    ref_q = O.codebook_lookup(idx.cpu().long(), synth_weights, 2, 16, 256)
    assert torch.equal(nchw(quant), ref_q)


def test_gmflow_vs_golden(gpu_net):
    a = synth.synth_clip(T=2, B=1, size=64, seed=99)[0]
    flow = gpu_net._gmflow(a[1:2].cuda(), a[0:1].cuda())
    close(nchw(flow), OPS['gmflow64'], 5e-4, 'gmflow 64x64')


def test_gmflow256_physical_regime_vs_reference_golden(gpu_net):
    """M16-M21 per module at a physical flow scale (tests/golden/gmflow256.npz: the reference's FlowGenerator on frames 0 and 3 of
    the translating texture at 256x256 -- |flow| median 0.87 px, p90 2.3, p99 6.9, max 88 px), EVERY pixel: within 2e-3 px
    (2e-5 of the flow scale), and the 99th percentile of the per-pixel error relative to max(1 px, |reference flow|) within 3e-4."""
    g = np.load(os.path.join(GOLDEN, 'gmflow256.npz'))
    dt = int(g['dt'])
    a = synth.synth_clip(T=dt + 1, B=1, size=256, seed=int(g['clip_seed']))[0]
    flow = nchw(gpu_net._gmflow(a[dt:dt + 1].cuda(), a[0:1].cuda())).numpy()
    ref = g['flow']
    err = np.abs(flow - ref)
    mag = np.maximum(1.0, np.sqrt((ref ** 2).sum(1, keepdims=True)))
    rep = {'max_err_px': float(err.max()), 'p99_rel': float(np.quantile(err / mag, 0.99)), 'rms_px': float(np.sqrt((err ** 2).mean())),
           'median_flow_px': float(np.median(np.sqrt((ref ** 2).sum(1)))), 'scale_px': float(np.abs(ref).max())}
    print(f'gmflow256 [{gpu_net.precision}]', rep)
    assert np.isfinite(flow).all() and 0.5 < rep['median_flow_px'] < 2.0
    # measured on MI355X (round 5): max 5.4e-4 / 4.5e-4 px (x3 / fp32), p99 relative 7.8e-5 / 8.2e-5, rms 4.3e-5 px
    assert rep['max_err_px'] <= 2e-5 * rep['scale_px'] and rep['max_err_px'] <= 2e-3, rep
    assert rep['p99_rel'] <= 3e-4, rep


def test_gmflow_clip_layer0_runs_once_per_frame(gpu_net, monkeypatch):
    """KeepNet._gmflow_clip: the position table and the self-attention block of GMFlow's layer 0 see one image and nothing of its
    pair, so the clip form runs them once per frame and gathers into pair order (an interior frame sits in two pairs).  Same bits
    as running them on every pair member (per-image arithmetic and plans), and as the pair-at-a-time entry point."""
    from comfyui_keep_amd.engine import net as net_mod
    x = synth.synth_clip(T=4, B=2, size=64, seed=7).cuda()
    monkeypatch.setattr(net_mod, 'GM_DEDUP_L0', True)
    f_once = gpu_net._gmflow_clip(x)
    monkeypatch.setattr(net_mod, 'GM_DEDUP_L0', False)
    f_pairs = gpu_net._gmflow_clip(x)
    assert f_once.shape == (2 * 3, 64, 64, 2) and torch.equal(f_once, f_pairs)
    solo = gpu_net._gmflow(x[1, 2:3], x[1, 1:2])                      # clip 1, pair (frame 2, frame 1)
    assert torch.equal(solo[0], f_once[3 + 1])


def test_encoder_and_generator_stacks_vs_golden(gpu_net):
    z, _ = gpu_net._vq_stack(nhwc(op_input('encoder64', (1, 3, 64, 64))), 'encoder', encoder_blocks(DEFAULT_ARCH))
    close(nchw(z), OPS['encoder64'], 5e-4, 'encoder 64x64')
    y, _ = gpu_net._vq_stack(nhwc(op_input('generator_2x2', (1, 256, 2, 2), 0.7)), 'generator', generator_blocks(DEFAULT_ARCH))
    close(nchw(y), OPS['generator_2x2'], 5e-4, 'generator from 2x2')


def _digest(frames):
    T, C, H, Wd = frames.shape
    return frames[:, :, 7::H // 32, 5::Wd // 32][:, :, :32, :32]


def _full_forward_check(net, gold_name, T, wide=False):
    """Free running (own GMFlow, own indices: NO injection) against the golden vectors of the imported reference.
    Frame 0 does not depend on the optical flow and is asserted strictly.  Later frames see the flow through the warped
    previous output; the recurrence is chaotic once ONE token flips (the next frame restores a different prev_out), so indices
    are compared frame by frame up to and including the first frame with a flip:
      * the top-1 logit of every token is within LOGIT_ERR_BOUND of the reference's (asserted, printed per frame);
      * every token whose reference margin exceeds max(1e-3, 2 x the MEASURED logit error of this run) agrees;
      * a flipped token has a margin below twice the measured logit error -- nowhere else;
      * the restored frames BEFORE the first flip are within 1e-3 max-abs of the reference, absolute.
    The arithmetic over all T frames is pinned separately with the reference's indices injected (<= 1e-3 absolute and
    <= 3e-4 of the output scale), and all-frames / all-pixels free running against the oracle with its flows injected (tests
    below).  (The reference goldens hold a 32 x 32 strided digest + per-channel statistics per frame, not every pixel.)
    wide: round 3's regime (flows of hundreds of pixels) -- the out-of-range edge case, with its own logit bound."""
    bound = LOGIT_ERR_BOUNDS.get(gold_name, LOGIT_ERR_BOUND_WIDE if wide else LOGIT_ERR_BOUND)
    assert bound <= (LOGIT_ERR_BOUND_WIDE if wide else LOGIT_ERR_BOUND)
    g = np.load(os.path.join(GOLDEN, gold_name))
    x = synth.synth_clip(T=T, B=1, seed=1234, pattern='waves' if wide else 'texture').cuda()
    out, aux = net(x, need_upscale=False, return_aux=True)
    idx = aux['indices'][0].cpu().numpy().astype(np.int16)
    agree = (idx == g['indices'])
    first_div = next((t for t in range(T) if not agree[t].all()), T)
    top1 = aux['logit_top1'][0].cpu().numpy()
    upto = min(first_div + 1, T)
    dlogit = np.abs(top1[:upto] - g['logit_top1'][:upto])
    dlogit_agree = float(dlogit[agree[:upto]].max())
    free = np.abs(_digest(out[0].cpu()).numpy() - g['out_grid']).reshape(T, -1).max(1)
    scale = float(np.abs(g['out_grid']).max())
    report = {'index_agreement': float(agree.mean()), 'frame0_agreement': float(agree[0].mean()),
              'first_frame_with_a_flip': first_div,
              'max_top1_logit_err_up_to_first_flip': dlogit_agree,
              'per_frame_top1_logit_err': [round(float(dlogit[t][agree[t]].max()), 6) for t in range(upto)],
              'gain_err': float(np.abs(aux['gains'][0].cpu().numpy() - g['gains']).max()),
              'flow_err_px': float(np.abs(_digest(aux['flows'][0].permute(0, 3, 1, 2).cpu()).numpy() - g['flow_grid']).max()),
              'flow_scale_px': float(np.abs(g['flow_grid']).max()),
              'flow_median_px': float(np.median(np.sqrt((g['flow_grid'] ** 2).sum(1)))),
              'margins_of_first_flips': (g['margins'][first_div][~agree[first_div]].tolist() if first_div < T else []),
              'free_running_pixel_err_before_first_flip': [round(float(v), 7) for v in free[:first_div]]}
    print(gold_name, f'[{net.precision}]', report)
    assert report['flow_err_px'] <= 2e-4 * max(1.0, report['flow_scale_px']), report
    assert report['gain_err'] <= 2e-4, report
    assert agree[0][g['margins'][0] > 1e-3].all(), report
    assert float(dlogit[0].max()) <= 1e-3, report                 # frame 0 sees no flow: fp32 re-association only
    assert dlogit_agree <= bound, report
    flip_margin = 2.0 * dlogit_agree                              # derived from THIS run's measured logit error
    for t in range(upto):
        assert agree[t][g['margins'][t] > max(1e-3, flip_margin)].all(), (t, report)
    assert first_div >= 1 and all(m <= flip_margin for m in report['margins_of_first_flips']), report
    if T <= 3:
        assert agree.mean() >= 0.99, report
    # free running, no injection: the frames before the first flip, absolute 1e-3 (north_star) and relative to the frames' scale
    if first_div > 0:
        assert float(free[:first_div].max()) <= min(1e-3, 3e-4 * scale), report
    assert float(free[0]) <= 5e-5 * scale, report                 # frame 0: fp32 re-association only
    # arithmetic drift with the reference's indices injected (separates index flips from drift)
    forced = torch.from_numpy(g['indices'].astype(np.int32)).view(1, T, -1)
    out_f = net(x, need_upscale=False, force_indices=forced)
    per_frame = np.abs(_digest(out_f[0].cpu()).numpy() - g['out_grid']).reshape(T, -1).max(1)
    err_f = float(per_frame.max())
    print(gold_name, f'[{net.precision}] max-abs pixel diff (reference indices injected): {err_f:.3e}; per frame:',
          [round(float(v), 6) for v in per_frame], f'; output scale {scale:.3g}')
    # <= 1e-3 max-abs (north_star), ABSOLUTE, at every clip length incl. the metric's own T = 20 (the synthetic net's frames live
    # in the range the tolerance is stated for: engine/synth.py HEAD_GAIN) AND relative to the frames' scale (a change of the
    # synthetic output gain must not loosen the gate)
    assert err_f <= 1e-3 and err_f <= 3e-4 * scale, (err_f, scale)
    st = out_f[0].cpu().reshape(T, 3, -1)
    stats = torch.stack([st.mean(-1), st.std(-1), st.min(-1).values, st.max(-1).values], -1).numpy()
    assert np.abs(stats - g['out_stats']).max() <= 2e-3
    return out


def test_full_forward_T3_vs_oracle_flows_injected(gpu_net, synth_weights):
    """All frames strict: the oracle's flows are injected so that the only differences left are fp32
    re-association in the conv / attention stacks; indices must agree wherever the oracle's margin > 1e-3 and the
    free-running output must be within 1e-3 max-abs of the oracle on EVERY pixel."""
    x = synth.synth_clip(T=3, B=1, seed=1234)
    ref, raux = O.keep_forward(x, synth_weights, return_aux=True)
    out, aux = gpu_net(x.cuda(), return_aux=True, force_flows=raux['flows'])
    top2 = raux['logits'].topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    agree = aux['indices'].cpu().long() == raux['indices']
    # frame by frame up to and including the first frame with ANY differing token (a sub-1e-3-margin flip there makes the
    # next frame restore a different prev_out: beyond it the two runs are not comparable token by token)
    first_div = next((t for t in range(3) if not bool(agree[0, t].all())), 3)
    print('safe fraction', safe.float().mean().item(), 'agreement', agree.float().mean().item(), 'first frame with a flip', first_div)
    for t in range(min(first_div + 1, 3)):
        assert agree[0, t][safe[0, t]].all(), f'frame {t}: a token with margin > 1e-3 differs'
    assert first_div >= 1
    assert (aux['gains'].cpu() - raux['gains'].view(1, 3, -1)).abs().max().item() <= 2e-4
    if first_div > 0:
        err0 = (out.cpu() - ref)[0, :first_div].abs().max().item()
        print('max-abs pixel diff, all pixels, frames before the first flip:', err0)
        assert err0 <= 1e-3
    if agree.all():
        err = (out.cpu() - ref).abs().max().item()
        print('max-abs pixel diff, all pixels, free running:', err)
        assert err <= 1e-3
    own = gpu_net(x.cuda(), return_aux=True)[1]['flows']            # the engine's own GMFlow vs the oracle's
    ferr = (own.permute(0, 1, 4, 2, 3).cpu() - raux['flows']).abs().max().item()
    print('flow max-abs diff (px):', ferr, 'flow scale', raux['flows'].abs().max().item())
    assert ferr <= 2e-4 * raux['flows'].abs().max().item()


def test_full_forward_T3_vs_reference_golden(gpu_net):
    out = _full_forward_check(gpu_net, 'keep_forward_T3.npz', 3)
    # every pixel of the 128x128 centre crop of all three frames against the REFERENCE itself (keep_forward_T3_pixels.npz,
    # oracle/make_golden_r5.py), free running: T = 3 agrees on every token (asserted above: agreement >= 0.99 and the flip rule),
    # so the frames are comparable pixel by pixel
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T3_pixels.npz'))
    a, b, c, d = (int(v) for v in g['crop'])
    err = np.abs(out[0][:, :, a:b, c:d].cpu().numpy() - g['out_crop']).reshape(3, -1).max(1)
    print(f'T3 every pixel of the centre crop vs the reference [{gpu_net.precision}], per frame:', err)
    assert float(err[0]) <= 5e-5 and float(err.max()) <= 2e-4, err       # measured 3.9e-6 ... 1.3e-5 (x3), 4.9e-6 ... 3.0e-5 (fp32)


def test_full_forward_T3_wide_flow_regime_vs_reference_golden(gpu_net):
    """Round 3's regime as the out-of-range edge case of the warp path: i.i.d. flownet weights + the plane-wave clip give flows
    of hundreds of pixels (median 97 px, 99.7 % above 8 px: most warp samples fall outside the frame)."""
    from comfyui_keep_amd.engine.net import KeepNet
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(synth.synth_state_dict(DEFAULT_ARCH, seed=0, flow_regime='wide'), strict=True)
    net.to('cuda').eval().set_precision(gpu_net.precision)
    _full_forward_check(net, 'keep_forward_T3_wide.npz', 3, wide=True)


def test_full_forward_T20_vs_reference_golden(gpu_net):
    """The metric's own clip length: 19 recurrent steps of prev_out -> warp -> hq_encoder -> Kalman -> indices
    (keep_arch.py:1062-1127) against the imported reference (tests/golden/keep_forward_T20.npz)."""
    out = _full_forward_check(gpu_net, 'keep_forward_T20.npz', 20)
    # round 6: EVERY pixel of the 128x128 centre crop of frames 0-3 against the REFERENCE itself at the metric's own clip length
    # (keep_forward_T20_pixels.npz, oracle/make_golden_r6.py), free running: the frames in front of the first token whose reference
    # margin (7.5e-4, frame 4) lies below the logit error -- _full_forward_check has just asserted that no index differs before it
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T20_pixels.npz'))
    a, b, c, d = (int(v) for v in g['crop'])
    frames = [int(v) for v in g['frames']]
    err = np.abs(out[0][frames][:, :, a:b, c:d].cpu().numpy() - g['out_crop']).reshape(len(frames), -1).max(1)
    print(f'T20 every pixel of the centre crop of frames {frames} vs the reference [{gpu_net.precision}]:', err)
    assert float(err[0]) <= 5e-5 and float(err.max()) <= 2e-4, err


def test_x3_vs_exact_f32_first_flips_are_explained_by_the_logit_difference(synth_weights):
    """What bench.py prints as `vs_exact_f32_policy`, asserted: the x3 policy against the exact-f32 policy on the same clips, free
    running.  Both are fp32-grade arithmetics of one graph; their code indices may part only where the exact-f32 run's own top-1 /
    top-2 margin is below twice the top-1 logit difference MEASURED between the two runs up to that frame (the flip rule of
    `_full_forward_check`, with the exact policy in the reference's place), frame 0 (no flow, no recurrence) must agree on every
    token above margin 1e-3, and the frames in front of a clip's first flip must be within 1e-3."""
    from comfyui_keep_amd.engine.net import KeepNet
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(synth_weights, strict=True)
    net.to('cuda').eval()
    B, T = 4, 20
    x = torch.cat([synth.synth_clip(T=T, B=1, seed=1234 + 7 * b, pattern='texture') for b in range(B)]).cuda()
    net.set_precision('fp32')
    ref, raux = net(x, return_aux=True)
    ref, rmargin, ridx, rtop = ref.cpu(), raux['margins'].cpu(), raux['indices'].cpu(), raux['logit_top1'].cpu()
    net.set_precision('x3')
    out, aux = net(x, return_aux=True)
    agree = aux['indices'].cpu() == ridx                              # [B,T,256]
    dl = (aux['logit_top1'].cpu() - rtop).abs()
    rows = []
    for b in range(B):
        first = next((t for t in range(T) if not bool(agree[b, t].all())), T)
        upto = min(first + 1, T)
        measured = float(dl[b, :upto][agree[b, :upto]].max())
        flips = rmargin[b, first][~agree[b, first]].tolist() if first < T else []
        before = float((out[b, :first].cpu() - ref[b, :first]).abs().max()) if first > 0 else 0.0
        rows.append((first, round(measured, 6), [round(m, 6) for m in flips], round(before, 7)))
        assert bool(agree[b, 0][rmargin[b, 0] > 1e-3].all()), rows
        assert first >= 1 and measured <= LOGIT_ERR_BOUNDS['keep_forward_T20.npz'], rows
        for t in range(upto):
            assert bool(agree[b, t][rmargin[b, t] > max(1e-3, 2.0 * measured)].all()), (b, t, rows)
        assert all(m < 2.0 * measured for m in flips), rows
        assert before <= 1e-3, rows
    print('x3 vs exact f32, per clip (first frame with a flip, measured top-1 logit difference, margins of the first flips, '
          'max-abs pixel difference before it):', rows)


def test_full_forward_T20_vs_oracle_drift_report(gpu_net, synth_weights):
    """T = 20, free running, the oracle's flows injected (so that GMFlow's 4096-way softmax re-association is out of the
    comparison): code indices must match the oracle on every token whose margin exceeds 1e-3, frame by frame, up to the
    first frame where any (low-margin) token differs -- beyond it the two runs restore different inputs.  If no token
    flips over the whole clip, every pixel of all 20 frames must be within 1e-3."""
    x = synth.synth_clip(T=20, B=1, seed=1234)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, raux = O.keep_forward(x, synth_weights, return_aux=True)
    out, aux = gpu_net(x.cuda(), return_aux=True, force_flows=raux['flows'])
    top2 = raux['logits'].topk(2, -1).values
    margin = (top2[..., 0] - top2[..., 1])[0]                      # [T, 256]
    agree = (aux['indices'].cpu().long() == raux['indices'])[0]    # [T, 256]
    first_div = next((t for t in range(20) if not bool(agree[t].all())), 20)
    gain_err = (aux['gains'].cpu() - raux['gains'].view(1, 20, -1)).abs().max().item()
    per_frame = (out.cpu() - ref)[0].abs().flatten(1).max(1).values
    print(f'T=20 [{gpu_net.precision}]: first frame with a differing index: {first_div} (20 = none); min margin '
          f'{margin.min().item():.2e}; gain err {gain_err:.2e}; per-frame max-abs pixel diff up to there: '
          f'{[round(float(v), 6) for v in per_frame[:max(first_div, 1)]]}')
    assert gain_err <= 2e-4
    for t in range(min(first_div + 1, 20)):
        assert agree[t][margin[t] > 1e-3].all(), f'frame {t}: a token with margin > 1e-3 differs'
    assert first_div >= 1
    assert float(per_frame[:first_div].max()) <= 1e-3, per_frame[:first_div]          # absolute, every pixel, every frame


def _stub_helper_pack(net):
    import test_host_logic as H   # installs the ComfyUI stubs
    from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
    pack = KEEPModelPack(net, H._Helper(), None, None, 'KEEP')
    pack.device = torch.device('cuda')
    return pack


@pytest.mark.parametrize("n_crops,faces", [(300, 1), (900, 3)])
def test_config3_config4_clip_mixes_equal_sequential(gpu_net, n_crops, faces):
    """BASELINE configs[2] / [3] as the hot path sees them (keep_processor.py:256-276): 300 crops of one face = 15 clips
    x 20, 900 frame-major interleaved crops of 3 faces = 45 clips x 20, handed to the engine in one call (batched on the
    batch axis by free HBM).  Order and chunking must equal the sequential one-clip-at-a-time loop: checked bit-exactly
    on the uint8 output for a spread of clips (restored one at a time through the same net) and structurally for all."""
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor, split_clips
    proc = KEEPFaceProcessor(_stub_helper_pack(gpu_net))
    base = synth.ramp_image()
    g = np.random.default_rng(n_crops)
    # crop k: frame k // faces, face k % faces -- distinct content per (frame, face)
    crops = [np.ascontiguousarray((np.roll(base, (7 * (k // faces)) % 512, axis=1).astype(np.int16)
                                   + 40 * (k % faces) + g.integers(-8, 9, (512, 512, 3))).clip(0, 255).astype(np.uint8))
             for k in range(n_crops)]
    faces_out = proc._restore_crops_u8(crops, 20)
    assert len(faces_out) == n_crops and all(f.shape == (512, 512, 3) and f.dtype == np.uint8 for f in faces_out)
    spans = split_clips(n_crops, 20)
    assert len(spans) == n_crops // 20 and all(e - s == 20 for s, e in spans)
    # Kernel tile / split-K choices follow the per-image geometry only (keep_conv2d_plan), so a clip's
    # frames do not depend on its batch-mates: every frame of every checked clip must equal the solo run BIT FOR BIT.
    firsts = gpu_net.run_clips_u8([torch.from_numpy(np.stack(crops[s:s + 1])) for s, _ in spans], max_b=16)
    for ci, (s, e) in enumerate(spans):
        assert np.array_equal(firsts[ci].numpy()[0], faces_out[s]), ('frame 0 of clip', ci)
    for ci in sorted({0, len(spans) // 2, len(spans) - 1}):
        s, e = spans[ci]
        solo = gpu_net.run_clips_u8([torch.from_numpy(np.stack(crops[s:e]))], max_b=1)[0].numpy()
        assert np.array_equal(solo, np.stack(faces_out[s:e])), ('clip', ci)


def test_full_forward_asian_T2_vs_reference_golden():
    from comfyui_keep_amd.engine.net import KeepNet
    cfg = dict(DEFAULT_ARCH, cft_list=['32', '64', '128', '256'], temp_reg_list=[])
    net = KeepNet(**cfg)
    net.load_state_dict(synth.synth_state_dict(cfg, seed=0), strict=True)
    _full_forward_check(net.to('cuda').eval(), 'keep_forward_asian_T2.npz', 2)


def test_hipgraph_replay_is_bit_identical_to_eager(gpu_net):
    """Small batches run as a captured hipGraph (one replay instead of ~9 k launches per clip): same kernels, same order,
    same pointers' worth of arithmetic -> bit-identical output, also on a second input through the same graph."""
    x1 = synth.synth_clip(T=3, B=1, seed=1234).cuda()
    x2 = synth.synth_clip(T=3, B=1, seed=99, phase=0.5).cuda()
    mode = gpu_net.graph_mode
    try:
        gpu_net.graph_mode = '0'
        e1, e2 = gpu_net(x1), gpu_net(x2)
        gpu_net.graph_mode = '1'
        g1 = gpu_net(x1)            # captures
        g2 = gpu_net(x2)            # replays the same graph on new input
        g1b = gpu_net(x1)
        assert any(k[:4] == (1, 3, 512, 512) for k in gpu_net._graphs)
        assert torch.equal(e1, g1) and torch.equal(e2, g2) and torch.equal(e1, g1b)
    finally:
        gpu_net.graph_mode = mode


def test_two_stream_order_of_the_forward_is_bit_identical(gpu_net, monkeypatch):
    """Round 5: with at most KEEP_AMD_OVERLAP_MAX_CLIPS clips in flight GMFlow (in chunks of pairs) and the Kalman gains run on a
    second stream under the frame recurrence.  Same kernels on the same data -- per-image arithmetic and plans, so chunking the pair
    batch changes nothing: output, flows, gains and indices equal the one-stream order BIT FOR BIT, eagerly and through a captured
    hipGraph (fork / join recorded), for one clip and for two, with chunk boundaries that do and do not divide T - 1."""
    from comfyui_keep_amd.engine import net as net_mod
    mode = gpu_net.graph_mode
    try:
        for B, T, first, chunk in ((1, 6, 3, 4), (2, 5, 1, 2), (1, 4, 2, 1)):
            x = synth.synth_clip(T=T, B=B, seed=77 + T).cuda()
            gpu_net.graph_mode = '0'
            monkeypatch.setattr(net_mod, 'STREAM_OVERLAP_MAX_CLIPS', 0)
            ref, raux = gpu_net(x, return_aux=True)
            monkeypatch.setattr(net_mod, 'STREAM_OVERLAP_MAX_CLIPS', 2)
            monkeypatch.setattr(net_mod, 'STREAM_OVERLAP_FIRST', first)
            monkeypatch.setattr(net_mod, 'STREAM_OVERLAP_CHUNK', chunk)
            got, aux = gpu_net(x, return_aux=True)
            assert torch.equal(aux['flows'], raux['flows']) and torch.equal(aux['gains'], raux['gains'])
            assert torch.equal(aux['indices'], raux['indices']) and torch.equal(got, ref)
            gpu_net.graph_mode = '1'
            g1 = gpu_net(x)            # captures the two-stream order
            g2 = gpu_net(x)            # replays it
            assert torch.equal(g1, ref) and torch.equal(g2, ref)
    finally:
        gpu_net.graph_mode = mode


def test_graph_survives_a_policy_switch_and_auto_captures_on_second_use(gpu_net):
    """(a) The per-forward bookkeeping block (status word + max|out| arena) lives as long as the net's Ops: a captured x3
    graph must replay correctly after forwards under another policy (round 2 freed the arena on a policy switch and the
    replay wrote through the dangling pointer).  (b) graph mode 'auto' runs the first occurrence of a shape eagerly and
    captures on the second, so a shape seen once never pays for a capture."""
    if gpu_net.precision != 'x3':
        pytest.skip('x3 bookkeeping')
    x = synth.synth_clip(T=2, B=1, seed=11).cuda()
    mode = gpu_net.graph_mode
    try:
        gpu_net.graph_mode = '0'
        eager = gpu_net(x)
        gpu_net.graph_mode = 'auto'
        gpu_net._graphs.clear()
        gpu_net._graph_seen.clear()
        a = gpu_net(x)
        assert not gpu_net._graphs                        # first occurrence: eager
        b = gpu_net(x)
        assert len(gpu_net._graphs) == 1                  # second: captured + replayed
        gpu_net.set_precision('fp32')
        f32 = gpu_net(x)                                  # another policy in between (own graph key, same Ops block)
        junk = [torch.randn(1 << 16, device='cuda') for _ in range(64)]   # allocator churn where the arena used to be freed
        gpu_net.set_precision('x3')
        c = gpu_net(x)                                    # replay of the x3 graph
        assert torch.equal(eager, a) and torch.equal(eager, b) and torch.equal(eager, c)
        assert torch.isfinite(f32).all() and len(junk) == 64
    finally:
        gpu_net.set_precision('x3')
        gpu_net.graph_mode = mode


def test_x3_overflow_on_the_index_chain_falls_back_to_f32(synth_weights):
    """an fp16-range overflow that reaches the output ONLY through the arg-max (hq_encoder / Kalman update
    / code transformer: every linear there is an un-probed `bounded` operand) used to select code 0 and paint a finite, wrong
    frame.  Here one transformer MLP weight is scaled so that gelu(linear1) leaves the fp16 range on the x3 kernels: the
    arg-max raises the status word, the batch is re-run on the exact-f32 kernels, and the result equals the fp32 policy's."""
    from comfyui_keep_amd.engine.net import KeepNet
    W = dict(synth_weights)
    W['ft_layers.4.linear1.weight'] = W['ft_layers.4.linear1.weight'] * 3.0e5
    W['ft_layers.4.linear2.weight'] = W['ft_layers.4.linear2.weight'] / 3.0e5      # the fp32 net stays O(1) downstream
    x = synth.synth_clip(T=2, B=1, seed=21).cuda()
    nets = {}
    for pol in ('x3', 'fp32'):
        n = KeepNet(**DEFAULT_ARCH)
        n.load_state_dict(W, strict=True)
        nets[pol] = n.to('cuda').eval().set_precision(pol)
    ref = nets['fp32'](x)
    assert torch.isfinite(ref).all() and nets['fp32'].x3_fallbacks == 0
    got = nets['x3'](x)
    assert nets['x3'].x3_fallbacks == 1
    assert torch.equal(got, ref)
    # the uint8 entry point (deferred check, two streams) takes the same decision
    u8 = [torch.randint(0, 256, (2, 512, 512, 3), dtype=torch.uint8)]
    r_x3 = nets['x3'].run_clips_u8(u8)[0]
    r_32 = nets['fp32'].run_clips_u8(u8)[0]
    assert nets['x3'].x3_fallbacks == 2 and torch.equal(r_x3, r_32)


def _two_rank_worker(rank, world, port, out_dir, n_clips):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), KEEP_DIST_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    load_package()
    from comfyui_keep_amd.engine import dist as kdist, synth as S
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH as ARCH
    from comfyui_keep_amd.engine.net import KeepNet
    kdist.init_from_env(backend='gloo')
    torch.cuda.set_device(0)
    net = KeepNet(**ARCH)
    if rank == 0:
        net.load_state_dict(S.synth_state_dict(seed=0), strict=True)
        net.to('cuda')
        index, blob = net._index, net.packed_blob()
    else:
        index, blob = None, None
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    index, blob = kdist.broadcast_packed_weights(index, blob, src=0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if rank != 0:
        assert blob.is_cuda
        net.adopt_packed(index, blob)
    net.eval()
    g = torch.Generator().manual_seed(5)
    clips = [torch.randint(0, 256, (2 if c % 3 else 1, 512, 512, 3), generator=g, dtype=torch.uint8) for c in range(n_clips)]
    res = net.run_clips_u8(clips, max_b=2)                         # sharded round-robin, tensor gather to rank 0
    if rank == 0:
        print(f'broadcast_ms {ms:.1f} for {blob.numel() * 4 / 1e6:.0f} MB ({world} ranks on one device, gloo wire)', flush=True)
        np.savez(os.path.join(out_dir, 'sharded.npz'), *[r.numpy() for r in res], broadcast_ms=ms)
    else:
        assert res is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_share_the_gpu_product_path(tmp_path, synth_weights):
    """The N > 1 PRODUCT path executed on a GPU: 2 processes share device 0, the real 633 MB packed blob goes through
    broadcast_packed_weights + adopt_packed, 6 ragged clips are sharded by run_clips_u8 and collected with the uint8
    tensor gather; the result must equal the single-process run bit for bit.  The wire here is gloo: RCCL refuses two ranks
    on one device ("Duplicate GPU detected", NCCL 2.26 ncclInvalidUsage -- tried, round 3), so the RCCL transport itself
    can only run where N GPUs exist (the driver's scaling run); the world-1 test below initialises it on this one."""
    import socket
    import torch.multiprocessing as mp
    from comfyui_keep_amd.engine.net import KeepNet
    with socket.socket() as sck:
        sck.bind(('127.0.0.1', 0))
        port = sck.getsockname()[1]
    n_clips = 6
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), n_clips), nprocs=2, join=True)
    got = np.load(tmp_path / 'sharded.npz')
    print('weight broadcast, 2 ranks on one device:', float(got['broadcast_ms']), 'ms')
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(synth_weights, strict=True)
    net.to('cuda').eval()
    g = torch.Generator().manual_seed(5)
    clips = [torch.randint(0, 256, (2 if c % 3 else 1, 512, 512, 3), generator=g, dtype=torch.uint8) for c in range(n_clips)]
    solo = net.run_clips_u8(clips, max_b=2)
    for c in range(n_clips):
        assert np.array_equal(got[f'arr_{c}'], solo[c].numpy()), c


def test_single_process_pool_drives_two_workers(synth_weights, monkeypatch):
    """What a ComfyUI node can reach (engine/pool.py): ONE process owns the weights, ``start_pool(3)`` spawns two worker processes
    (here on the same device: KEEP_DIST_DEVICE, gloo wire for the one weight broadcast), and ``run_clips_u8`` -- the call
    KEEPFaceProcessor makes -- shards 7 ragged clips over root + workers.  Bit-equal to the same net without the pool, through the
    tensor entry point and through the processor's list-of-crops entry point; the process group does not outlive the broadcast."""
    from comfyui_keep_amd.engine.net import KeepNet
    monkeypatch.setenv('KEEP_DIST_DEVICE', '0')
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(synth_weights, strict=True)
    net.to('cuda').eval()
    g = torch.Generator().manual_seed(11)
    clips = [torch.randint(0, 256, (3 if c % 3 == 0 else (2 if c % 3 == 1 else 1), 512, 512, 3), generator=g, dtype=torch.uint8)
             for c in range(7)]
    solo = net.run_clips_u8(clips, max_b=2)
    pool = net.start_pool(3)
    try:
        assert not torch.distributed.is_initialized() and len(pool._procs) == 2
        print(f'pool of 3 on one device: weight broadcast {pool.broadcast_ms:.0f} ms')
        got = net.run_clips_u8(clips, max_b=2)
        assert len(got) == 7 and all(torch.equal(a, b) for a, b in zip(got, solo))
        as_lists = net.run_clips_u8([[f.numpy() for f in c] for c in clips], max_b=2)       # what _restore_crops_u8 hands over
        assert all(torch.equal(a, b) for a, b in zip(as_lists, solo))
        assert torch.equal(net.run_clips_u8(clips[:1])[0], solo[0])                        # a single clip stays on the root
    finally:
        pool.close()
        net.pool = None
    assert all(p.poll() is not None for p in pool._procs) or not pool._procs


def test_node_level_pool_from_the_environment(synth_weights, monkeypatch):
    """KEEP_AMD_GPUS=2 in the environment of the ONE ComfyUI process: ``KEEPModelPack.load_device()`` (keep_model_loader.py:28-43)
    starts the pool, ``KEEPFaceProcessor._restore_crops_u8`` -- the clip loop of keep_processor.py:263-270 -- shards its clips over
    root + worker; the restored crops equal the same processor without a pool, bit for bit."""
    import test_host_logic as H   # installs the ComfyUI stubs
    from comfyui_keep_amd.engine.net import KeepNet
    from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
    g = np.random.default_rng(3)
    crops = [g.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(7)]          # max_clip_length 2 -> clips of 2, 2, 2, 1

    def restored(with_pool):
        net = KeepNet(**DEFAULT_ARCH)
        net.load_state_dict(synth_weights, strict=True)
        pack = KEEPModelPack(net.eval(), H._Helper(), None, None, 'KEEP')
        pack.device = torch.device('cuda')
        pack.load_device()
        assert (net.pool is not None) == with_pool
        try:
            return KEEPFaceProcessor(pack)._restore_crops_u8(crops, 2)
        finally:
            if net.pool is not None:
                net.pool.close()
                net.pool = None

    solo = restored(False)
    monkeypatch.setenv('KEEP_AMD_GPUS', '2')
    monkeypatch.setenv('KEEP_DIST_DEVICE', '0')
    pooled = restored(True)
    assert len(pooled) == 7 and all(np.array_equal(a, b) for a, b in zip(pooled, solo))
    assert not torch.distributed.is_initialized()


def _rccl_world1_worker(rank, port, out_dir):
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import time
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend='nccl', rank=0, world_size=1)
    blob = torch.arange(1 << 24, dtype=torch.float32, device='cuda')          # 64 MB through ncclBroadcast / ncclAllGather
    t0 = time.perf_counter()
    torch.distributed.broadcast(blob, src=0)
    u8 = torch.full((1 << 20,), 7, dtype=torch.uint8, device='cuda')
    out = [torch.empty_like(u8)]
    torch.distributed.all_gather(out, u8)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    ok = bool(torch.equal(out[0], u8)) and float(blob[12345]) == 12345.0
    open(os.path.join(out_dir, 'rccl.txt'), 'w').write(f'{int(ok)} {ms:.2f}')
    torch.distributed.destroy_process_group()


def test_rccl_initialises_on_this_device(tmp_path):
    """backend 'nccl' IS RCCL on ROCm: communicator creation + one broadcast + one uint8 all_gather at world size 1."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sck:
        sck.bind(('127.0.0.1', 0))
        port = sck.getsockname()[1]
    mp.spawn(_rccl_world1_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    ok, ms = open(tmp_path / 'rccl.txt').read().split()
    print('RCCL world-1 broadcast + all_gather:', ms, 'ms')
    assert ok == '1'


def test_batched_clips_equal_sequential(gpu_net):
    """Independent clips on the batch axis give the same result as one at a time (hot loop #1 semantics)."""
    x = torch.cat([synth.synth_clip(T=2, B=1, seed=1234), synth.synth_clip(T=2, B=1, seed=77, phase=1.0)], 0).cuda()
    both, aux = gpu_net(x, return_aux=True)
    for b in range(2):
        one, aux1 = gpu_net(x[b:b + 1], return_aux=True)
        assert torch.equal(aux1['indices'][0], aux['indices'][b])
        assert torch.equal(one[0], both[b])        # plans follow the per-image geometry: no dependence on batch-mates
    outs = gpu_net.run_clips([x[0:1], x[1:2]])
    assert torch.equal(outs[1], both[1:2])
    # ... also across a very different batch (5 clips, the two above among them) and under graph replay of the solo run
    x5 = torch.cat([x, synth.synth_clip(T=2, B=3, seed=5, phase=0.3).cuda()], 0)
    five = gpu_net(x5)
    assert torch.equal(five[:2], both)


def test_processor_runs_on_engine(gpu_net):
    """Config 1 (BASELINE.json configs[0]) through the drop-in processor: one aligned 512x512 face (restored as a T=1
    clip by the engine; the reference's T=2 duplicate gives the same frame 0, see the single-frame test below)."""
    import test_host_logic as H   # installs the ComfyUI stubs
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
    from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
    from comfyui_keep_amd.modules import utils as U
    pack = KEEPModelPack(gpu_net, H._Helper(), None, None, 'KEEP')
    pack.device = torch.device('cuda')
    img = synth.ramp_image()
    out = KEEPFaceProcessor(pack).process_image(img, 1.0, True, True, False)
    assert out.shape == (512, 512, 3) and out.dtype == np.uint8
    x = U.crops_to_net_input([img]).unsqueeze(0).cuda()
    ref = gpu_net(x)[:, 0]
    assert np.array_equal(out, U.net_output_to_bgr_u8(ref[0]))


def test_processor_device_side_u8_path_equals_host_converters(gpu_net):
    """crops go to the GPU as uint8 and come back as uint8 (keep_img2tensor / keep_tensor2img either side
    of the clip loop); the result must be bit-identical to the reference's host converters around the same net, for a
    ragged chunking (3 crops, max_clip_length 2 -> clips of T=2 and T=1 -> duplicated)."""
    import test_host_logic as H   # installs the ComfyUI stubs
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
    from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
    from comfyui_keep_amd.modules import utils as U
    pack = KEEPModelPack(gpu_net, H._Helper(), None, None, 'KEEP')
    pack.device = torch.device('cuda')
    proc = KEEPFaceProcessor(pack)
    base = synth.ramp_image()
    crops = [np.ascontiguousarray(np.roll(base, 17 * k, axis=1)) for k in range(3)]
    dev_faces = proc._restore_crops_u8(crops, 2)

    class HostOnly:                      # same net, without the device-side converters
        supports_single_frame = True

        def __init__(self, net):
            self.net = net

        def run_clips(self, clips, need_upscale=False):
            return self.net.run_clips(clips, need_upscale=need_upscale)

    proc.keep_net = HostOnly(gpu_net)
    host_faces = proc._restore_crops_u8(crops, 2)
    assert len(dev_faces) == len(host_faces) == 3
    for a, b in zip(dev_faces, host_faces):
        assert a.shape == (512, 512, 3) and a.dtype == np.uint8
        assert np.array_equal(a, b)


def test_single_frame_fast_path_equals_frame0_of_duplicate(gpu_net):
    """Frame 0 of a clip depends on no other frame (no flow, no Kalman update, no CFA at i == 0): the engine restores a
    lone crop as T = 1 and must reproduce frame 0 of the reference's T = 2 duplicate (keep_processor.py:173-178)."""
    x = synth.synth_clip(T=1, B=2, seed=77).cuda()
    one = gpu_net(x)
    two = gpu_net(torch.cat([x, x], dim=1))
    assert one.shape == (2, 1, 3, 512, 512)
    assert torch.equal(one[:, 0], two[:, 0])          # same per-image plans at N = 2 and N = 4 frames
    import test_host_logic as H   # installs the ComfyUI stubs
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
    from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
    pack = KEEPModelPack(gpu_net, H._Helper(), None, None, 'KEEP')
    pack.device = torch.device('cuda')
    proc = KEEPFaceProcessor(pack)
    crop = [synth.ramp_image()]
    fast = proc._restore_crops_u8(crop, 20)

    class NoT1:
        supports_single_frame = False

        def __init__(self, net):
            self.run_clips_u8 = net.run_clips_u8

    proc.keep_net = NoT1(gpu_net)
    dup = proc._restore_crops_u8(crop, 20)
    assert np.array_equal(fast[0], dup[0])


def test_bf16_policy_quality_report(gpu_net):
    """bf16-MFMA policy (conv / linear operands rounded to bf16, fp32 accumulate and storage): measured against the
    reference golden -- index agreement and max-abs pixel error with the reference indices injected.  Code indices
    are an argmax over 1024 logits, so bf16 operand rounding (2^-9 relative) flips low-margin tokens: reported, and
    bounded loosely here; the <=1e-3 bound is the fp32 policy's."""
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T3.npz'))
    x = synth.synth_clip(T=3, B=1, seed=1234).cuda()
    own = gpu_net.precision
    gpu_net.set_precision('bf16')
    try:
        out, aux = gpu_net(x, return_aux=True)
        idx = aux['indices'][0].cpu().numpy().astype(np.int16)
        agree = idx == g['indices']
        forced = torch.from_numpy(g['indices'].astype(np.int32)).view(1, 3, -1)
        out_f = gpu_net(x, force_indices=forced)
        err_f = np.abs(_digest(out_f[0].cpu()).numpy() - g['out_grid']).max()
        gain_err = np.abs(aux['gains'][0].cpu().numpy() - g['gains']).max()
        print('bf16 policy: index agreement', agree.mean(), 'frame0', agree[0].mean(), 'confident(>0.5)',
              agree[g['margins'] > 0.5].mean(), 'gain err', gain_err, 'max-abs pixel diff (ref indices):', err_f)
        assert torch.isfinite(out).all()
        # frame 0 is flow-free; later frames are chaotic under the synthetic weights (see _full_forward_check)
        assert agree[0].mean() >= 0.9 and agree[0][g['margins'][0] > 0.5].mean() >= 0.97
        assert err_f <= 0.5 and gain_err <= 0.05
    finally:
        gpu_net.set_precision(own)
        gpu_net._activate_precision()


def test_need_upscale_runs_on_the_device(gpu_net):
    """K0: need_upscale=True (keep_arch.py:1020-1023) = x4 bilinear on the device, then the normal forward."""
    x = synth.synth_clip(T=1, B=1, size=128, seed=3).cuda()
    up = torch.nn.functional.interpolate(x.flatten(0, 1), scale_factor=4, mode='bilinear').unflatten(0, (1, 1))
    a = gpu_net(x, need_upscale=True)
    b = gpu_net(up.contiguous(), need_upscale=False)
    assert a.shape == (1, 1, 3, 512, 512) and (a - b).abs().max().item() <= 5e-4     # (keep_bilinear_upscale vs torch's interpolate)


def test_weights_stay_resident_across_offload(monkeypatch, synth_weights):
    """Residency policy: with KEEP_AMD_RESIDENT=1 the pack's offload() parks the packed weights on the device --
    the next load_device() neither uploads nor re-derives the policy's weight twin."""
    from comfyui_keep_amd.engine import net as netmod
    from comfyui_keep_amd.engine.net import KeepNet
    monkeypatch.setattr(netmod, 'RESIDENT', True)
    n = KeepNet(**DEFAULT_ARCH)
    n.load_state_dict(synth_weights, strict=True)
    n.to('cuda').eval()
    x = synth.synth_clip(T=1, B=1, seed=9).cuda()
    a = n(x)
    blob_ptr, twin = n._dev_blob.data_ptr(), n._dev_blobx3
    n.to('cpu')                                   # what KEEPModelPack.offload() does after every node call
    assert n._dev_blob is not None and n._dev_blob.data_ptr() == blob_ptr
    n.to('cuda')
    assert n._dev_blob.data_ptr() == blob_ptr and n._dev_blobx3 is twin
    assert torch.equal(n(x), a)
    monkeypatch.setattr(netmod, 'RESIDENT', False)
    n.to('cpu')
    assert n._dev_blob is None and n.w is None


def test_bench_multi_rank_line_shape(tmp_path):
    """the driver's SCALE run must not fail on plumbing.  `python bench.py --gpus 2` (self-launched under
    torch.distributed.run on the loopback address; KEEP_DIST_DEVICE=0 puts both ranks on this box's one GPU, gloo wire) prints ONE JSON
    line with the contract's keys and the multi-rank extras (broadcast_ms, frames_per_s_per_rank, config5_one_video_per_gpu)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KEEP_DIST_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--clips', '2',
                        '--no-extras', '--no-cpu-baseline'], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'broadcast_ms', 'broadcast_mb', 'frames_per_s_per_rank', 'config5_one_video_per_gpu'):
        assert k in line, k
    assert line['n_gpus'] == 2 and line['steps'] == 1 and line['warmup'] == 1 and line['scaling'] == 'weak' and line['higher_is_better'] is True
    assert line['dtype'] == 'f16x3' and line['data'] == 'synthetic' and line['vs_baseline'] is None and line['unit'] == 'frames/s'
    assert line['config']['clips_per_gpu'] == 2 and line['config']['parallelism'] == 'dp2 over clips' and 'workload' in line['config']
    assert len(line['frames_per_s_per_rank']) == 2 and all(v > 0 for v in line['frames_per_s_per_rank'])
    assert abs(line['value'] - 2 * 2 * 20 / (line['ms_per_step'] * 1e-3)) <= 0.01 * line['value']        # whole-job frames / max-over-ranks time
    assert line['broadcast_ms'] > 0 and 600 < line['broadcast_mb'] < 700
    c5 = line['config5_one_video_per_gpu']
    assert c5['crops_per_gpu'] == 300 and c5['clips_per_gpu'] == 15 and c5['value'] > 0
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in line['roofline'], k
