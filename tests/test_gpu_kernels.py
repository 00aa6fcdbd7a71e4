"""GPU suite (-m gpu): every libkeep_hip.so kernel, called through the C-ABI, against the CPU oracle
(torch fp32 ops / oracle/keep_oracle.py) on the same seeded inputs.  fp32-in / fp32-accumulate MFMA
path: tolerance 2e-4 relative to the output scale per op (re-association only), indices bit-exact.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import keep_oracle as O
from conftest import GOLDEN, op_input
from comfyui_keep_amd.engine import hiplib as L
from comfyui_keep_amd.engine import ops, synth

pytestmark = pytest.mark.gpu
TOL = 2e-4


def dev(t):
    return t.cuda().contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def check(got, ref, tol=TOL, what=''):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f'{what}: non-finite output'
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, f'{what}: max abs err {err:.3e} (scale {scale:.3g})'


def pack(w):
    return dev(w.permute(0, 2, 3, 1))


def rnd(name, shape, scale=1.0):
    return op_input(name, shape, scale)


# ------------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("cin,cout,hw,n", [(64, 64, 32, 2), (128, 256, 16, 1), (32, 48, 24, 3), (256, 128, 64, 1)])
def test_conv3x3_plain(cin, cout, hw, n):
    x, w, b = rnd('cx', (n, cin, hw, hw)), rnd('cw', (cout, cin, 3, 3), 0.05), rnd('cb', (cout,))
    y = ops.conv(dev(nhwc(x)), pack(w), dev(b), split_k=1)
    check(nchw(y), F.conv2d(x, w, b, padding=1), what='conv3x3')


def test_conv3x3_f32_halo_kernel_vs_gather_kernel(monkeypatch):
    """fp32 policy: the persistent LDS-halo kernel (default for 3x3 stride-1) against the implicit-GEMM gather kernel
    and against F.conv2d -- wide (8x32) and square (16x16) tiles, masked half cout-block, fused GN+swish, upsample."""
    for (n, cin, cout, h, wd, up) in [(2, 64, 64, 64, 64, False), (1, 128, 96, 16, 48, False), (2, 32, 128, 16, 16, True)]:
        x, w, b = rnd('hfx', (n, cin, h, wd), 2.0) + 0.3, rnd('hfw', (cout, cin, 3, 3), 0.05), rnd('hfb', (cout,))
        gamma, beta = rnd('hfg', (cin,)) * 0.2 + 1, rnd('hfbt', (cin,)) * 0.2
        xd = dev(nhwc(x))
        pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
        kw = dict(pro=pro, pro_act=L.PRO_SWISH, upsample=up, stats=True, split_k=1)
        y, st = ops.conv(xd, pack(w), dev(b), **kw)
        assert st is not None and st.P == (y.shape[1] * y.shape[2]) // 64
        ops.DEFAULT.flags = L.CONV_NO_HALO_F32
        y_g, _ = ops.conv(xd, pack(w), dev(b), **kw)
        ops.DEFAULT.flags = 0
        hn = F.group_norm(x, 32, gamma, beta, eps=1e-6)
        hn = hn * torch.sigmoid(hn)
        if up:
            hn = F.interpolate(hn, scale_factor=2.0, mode='nearest')
        ref = F.conv2d(hn, w, b, padding=1)
        check(nchw(y), ref, what='halo f32 vs torch')
        check(y, y_g, 2e-5, what='halo f32 vs gather kernel')
        sc, sh = ops.norm_affine(y, None, None, cout, 1e-5, stats=st)
        sc2, sh2 = ops.norm_affine(y, None, None, cout, 1e-5)
        check(sc, sc2, 1e-5, 'halo f32 fused stats scale'); check(sh, sh2, 1e-5, 'halo f32 fused stats shift')


# ------------------------------------------------------------------------------------------------ split fp16 (x3) policy
def x3w(wp):
    """(weight_x3, acc_scale) for a packed [Cout,KH,KW,Cin] weight: the host-side split of engine/ops.py."""
    sc = ops.x3_scale_for(float(wp.abs().max()))
    return ops.split_x3(wp.reshape(-1, wp.shape[-1]), sc).view(-1), 1.0 / sc


def err64(got, ref64):
    return (got.detach().double().cpu() - ref64).abs().max().item()


@pytest.mark.parametrize("n,cin,cout,h,wd,up,res,gn", [(2, 64, 64, 64, 64, False, True, True), (1, 128, 96, 16, 48, False, False, True),
                                                      (2, 32, 128, 16, 16, True, False, True), (1, 48, 64, 32, 32, False, True, False),
                                                      (1, 512, 512, 16, 16, False, False, True)])
def test_conv_x3_halo_is_fp32_grade(n, cin, cout, h, wd, up, res, gn):
    """KEEP_MMA_X3 3x3 halo kernel: the error against an fp64 convolution is of the size of the exact-f32 kernel's own
    (accumulation-order) error -- wide / square tiles, masked half cout-block, fused GN+swish, upsample, residual,
    auto split-K on the small map, Cin = 48 (three 16-channel chunks)."""
    x, w, b = rnd('x3x', (n, cin, h, wd), 2.0) + 0.3, rnd('x3w', (cout, cin, 3, 3), 0.05), rnd('x3b', (cout,))
    gamma, beta = rnd('x3g', (cin,)) * 0.2 + 1, rnd('x3bt', (cin,)) * 0.2
    Ho, Wo = (2 * h, 2 * wd) if up else (h, wd)
    r = rnd('x3r', (n, cout, Ho, Wo)) if res else None
    xd, wp = dev(nhwc(x)), pack(w)
    wx3, asc = x3w(wp)
    kw = dict(upsample=up, stats=True, residual=None if r is None else dev(nhwc(r)))
    if gn:
        kw.update(pro=ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6), pro_act=L.PRO_SWISH)
    ops.DEFAULT.profile = []
    y, st = ops.conv(xd, wp, dev(b), mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, **kw)
    kname = ops.DEFAULT.profile[-1][0]
    ops.DEFAULT.profile = None
    assert kname.startswith(('conv3x3_halo_x3_kernel', 'conv3x3_halo_x3s_kernel')), kname
    y32, _ = ops.conv(xd, wp, dev(b), **kw)
    hn = x.double()
    if gn:
        hn = F.group_norm(hn, 32, gamma.double(), beta.double(), eps=1e-6)
        hn = hn * torch.sigmoid(hn)
    if up:
        hn = F.interpolate(hn, scale_factor=2.0, mode='nearest')
    ref = F.conv2d(hn, w.double(), b.double(), padding=1)
    if r is not None:
        ref = ref + r.double()
    ref = ref.permute(0, 2, 3, 1)
    e3, e32 = err64(y, ref), err64(y32, ref)
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(y).all()
    assert e3 <= max(3.0 * e32, 2e-6 * scale), f'x3 err {e3:.3e} vs f32-kernel err {e32:.3e} (scale {scale:.3g})'
    if st is not None and st.amax is not None:     # epilogue-fused range probe of the output == a direct reduction
        assert torch.equal(st.amax.cpu(), y.abs().flatten(1).max(1).values.cpu())
    if st is not None and st.part is not None:
        sc, sh = ops.norm_affine(y, None, None, cout, 1e-5, stats=st)
        sc2, sh2 = ops.norm_affine(y, None, None, cout, 1e-5)
        check(sc, sc2, 1e-5, 'x3 halo fused stats scale'); check(sh, sh2, 1e-5, 'x3 halo fused stats shift')


def test_conv_x3_gather_is_fp32_grade():
    """KEEP_MMA_X3 gather kernel (token GEMMs, 1x1 / strided convs): fp32-grade against fp64; bias + GELU, residual,
    per-image GroupNorm prologue, Cin = 48 / 80 (half-empty last 32-channel step), stride-2 Downsample geometry, split-K."""
    def run(name, x4, w4, b, ref64, **kw):
        wp = pack(w4)
        wx3, asc = x3w(wp)
        ops.DEFAULT.profile = []
        y, st = ops.conv(dev(nhwc(x4)), wp, None if b is None else dev(b), mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, stats=True, **kw)
        assert ops.DEFAULT.profile[-1][0].startswith('conv_x3_kernel'), (name, ops.DEFAULT.profile[-1][0])
        ops.DEFAULT.profile = None
        if st is not None and st.amax is not None:
            assert torch.equal(st.amax.cpu(), y.abs().flatten(1).max(1).values.cpu()), name
        y32 = ops.conv(dev(nhwc(x4)), wp, None if b is None else dev(b), **kw)
        e3, e32 = err64(nchw(y), ref64), err64(nchw(y32), ref64)
        scale = max(1.0, ref64.abs().max().item())
        assert torch.isfinite(y).all()
        assert e3 <= max(3.0 * e32, 2e-6 * scale), f'{name}: x3 err {e3:.3e} vs f32-kernel err {e32:.3e} (scale {scale:.3g})'
    # token GEMM 512 -> 1024 with bias + exact GELU (code transformer MLP)
    x, w, b = rnd('g3x', (1, 512, 300, 1), 2.0), rnd('g3w', (1024, 512, 1, 1), 0.05), rnd('g3b', (1024,))
    run('linear+gelu', x, w, b, F.gelu(F.conv2d(x.double(), w.double(), b.double())), pad=0, ksize=1, act=L.ACT_GELU)
    # Downsample (pad right/bottom, stride 2), Cin = 80
    x, w, b = rnd('g3dx', (2, 80, 32, 32)), rnd('g3dw', (96, 80, 3, 3), 0.05), rnd('g3db', (96,))
    run('down', x, w, b, F.conv2d(F.pad(x.double(), (0, 1, 0, 1)), w.double(), b.double(), stride=2), down=True)
    # 1x1 conv with a per-image GroupNorm prologue (AttnBlock qkv) and a residual, Cin = 48, forced split-K
    x, w = rnd('g3px', (3, 48, 16, 16), 2.0) - 0.2, rnd('g3pw', (192, 48, 1, 1), 0.1)
    gamma, beta = rnd('g3pg', (48,)) * 0.2 + 1, rnd('g3pb', (48,)) * 0.2
    r = rnd('g3pr', (3, 192, 16, 16))
    xd = dev(nhwc(x))
    pro = ops.norm_affine(xd, dev(gamma), dev(beta), 16, 1e-6)
    ref = F.conv2d(F.group_norm(x.double(), 16, gamma.double(), beta.double(), eps=1e-6), w.double()) + r.double()
    run('1x1 gn', x, w, None, ref, pad=0, ksize=1, pro=pro, residual=dev(nhwc(r)))
    x, w = rnd('g3sx', (1, 256, 8, 8)), rnd('g3sw', (128, 256, 3, 3), 0.03)
    run('3x3 s2 split-k', x, w, None, F.conv2d(x.double(), w.double(), stride=2, padding=1), stride=2, pad=1, split_k=5)


def test_x3_subnormal_lo():
    """The x3 scheme relies on v_mfma_f32_32x32x16_f16 keeping SUBNORMAL fp16 inputs (gfx90a flushed them): activations of
    ~5e-3 have lo = x - fp16(x) ~ 1e-6, deep in the fp16 subnormal range.  Flushed lo terms would leave a 2^-12 = 2.4e-4
    relative error; kept, the error is the 2^-25 absolute floor of a subnormal lo (~1e-5 relative here)."""
    g = torch.Generator().manual_seed(5)
    x = (torch.rand((1, 64, 16, 32), generator=g) * 4e-3 + 3e-3)
    w = torch.ones((64, 64, 3, 3)) * 0.5                       # exactly representable: w_lo = 0, only a_lo * w_hi matters
    wp = pack(w)
    wx3 = ops.split_x3(wp.reshape(-1, 64), 1.0).view(-1)
    y = ops.conv(dev(nhwc(x)), wp, None, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=1.0)
    ref = F.conv2d(x.double(), w.double(), padding=1).permute(0, 2, 3, 1)
    rel = ((y.double().cpu() - ref).abs() / ref.abs()).max().item()
    assert rel < 2e-5, f'relative error {rel:.3e}: fp16 subnormals are being flushed by the matrix pipe'


def test_x3_overflow_is_loud():
    """An activation beyond the fp16 range does not produce a silently wrong value: the output is non-finite (the engine
    checks and re-runs the clip on the exact-f32 kernels)."""
    x = torch.ones((1, 32, 16, 16)) * 1e5
    w = rnd('ovw', (32, 32, 3, 3), 0.05)
    wp = pack(w)
    wx3, asc = x3w(wp)
    y = ops.conv(dev(nhwc(x)), wp, None, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, bounded=True)
    assert not torch.isfinite(y).all()


def test_x3_range_probe_keeps_large_raw_inputs_exact():
    """Un-normalised inputs (the raw residual stream in front of Upsample / Downsample / shortcut / CFT convs) are range-probed
    (keep_absmax) and rescaled per image by a power of two into the fp16 window: values far beyond 65504 and tiny ones
    both come out fp32-grade, per image (image 0 large, image 1 tiny), on the halo and on the gather kernel."""
    x = rnd('rpx', (2, 64, 32, 32), 1.0)
    x[0] *= 3e5
    x[1] *= 2e-4
    for name, w4, kw, ref in [
        ('halo', rnd('rpw', (64, 64, 3, 3), 0.05), dict(), lambda w: F.conv2d(x.double(), w.double(), padding=1)),
        ('gather', rnd('rpw1', (96, 64, 1, 1), 0.1), dict(pad=0, ksize=1), lambda w: F.conv2d(x.double(), w.double())),
    ]:
        wp = pack(w4)
        wx3, asc = x3w(wp)
        y = ops.conv(dev(nhwc(x)), wp, None, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, **kw)
        y32 = ops.conv(dev(nhwc(x)), wp, None, **kw)
        r64 = ref(w4)
        assert torch.isfinite(y).all(), name
        for n in range(2):
            e3 = err64(nchw(y)[n], r64[n])
            e32 = err64(nchw(y32)[n], r64[n])
            sc = r64[n].abs().max().item()
            assert e3 <= max(3.0 * e32, 2e-6 * sc), f'{name} image {n}: x3 err {e3:.3e} vs f32-kernel err {e32:.3e} (scale {sc:.3g})'


def test_attention_x3_is_fp32_grade_and_range_probed():
    """keep_attention KEEP_MMA_X3: error against an fp64 softmax(QK^T)V of the size of the exact-f32 kernel's own error; with
    `probe=True` operands of magnitude 1e5 (CFA: projections of the raw residual stream) stay finite and exact."""
    for (B, H, Lq, Lk, D, Dv, amp) in [(2, 8, 256, 256, 64, 64, 1.0), (3, 8, 20, 20, 48, 48, 1.0), (1, 1, 256, 256, 512, 512, 1.0),
                                       (2, 4, 200, 200, 256, 256, 1.0), (2, 4, 256, 256, 256, 256, 1e5), (2, 1, 1024, 1024, 128, 128, 1.0)]:
        q, k, v = rnd('xq', (B, Lq, H, D)) * amp, rnd('xk', (B, Lk, H, D)), rnd('xv', (B, Lk, H, Dv)) * amp
        scale = D ** -0.5 * 3.0 / amp
        ref = torch.softmax(torch.einsum('bqhd,bkhd->bhqk', q.double(), k.double()) * scale, -1)
        ref = torch.einsum('bhqk,bkhd->bqhd', ref, v.double())
        outs = []
        for mma in (L.MMA_X3, L.MMA_F32):
            o = torch.empty((B, Lq, H, Dv), device='cuda')
            ops.attention(dev(q), dev(k), dev(v), o, B=B, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=scale,
                          q_str=(Lq * H * D, H * D, D), k_str=(Lk * H * D, H * D, D), v_str=(Lk * H * Dv, H * Dv, Dv),
                          o_str=(Lq * H * Dv, H * Dv, Dv), mma=mma, probe=amp > 1)
            assert torch.isfinite(o).all()
            outs.append(err64(o, ref))
        sc = ref.abs().max().item()
        assert outs[0] <= max(3.0 * outs[1], 2e-6 * sc), f'{(B, H, Lq, Lk, D, Dv, amp)}: x3 err {outs[0]:.3e} vs f32 err {outs[1]:.3e} (scale {sc:.3g})'


def test_conv3x3_small_cout_valu_kernel(monkeypatch):
    """VQ:241 output conv (GroupNorm -> conv 64 -> 3): the Cout <= 4 fp32 VALU kernel vs torch and vs the gather kernel."""
    for cout in (3, 1, 4):
        x, w, b = rnd('scx', (2, 64, 16, 64), 2.0) - 0.4, rnd('scw', (cout, 64, 3, 3), 0.05), rnd('scb', (cout,))
        gamma, beta = rnd('scg', (64,)) * 0.2 + 1, rnd('scbt', (64,)) * 0.2
        xd = dev(nhwc(x))
        pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
        y = ops.conv(xd, pack(w), dev(b), pro=pro)
        ops.DEFAULT.flags = L.CONV_NO_COUT4
        y_g = ops.conv(xd, pack(w), dev(b), pro=pro)
        ops.DEFAULT.flags = 0
        ref = F.conv2d(F.group_norm(x, 32, gamma, beta, eps=1e-6), w, b, padding=1)
        check(nchw(y), ref, what=f'cout{cout} valu vs torch')
        check(y, y_g, 2e-5, what=f'cout{cout} valu vs gather kernel')
    # swish prologue + sigmoid epilogue, no bias, bf16 policy routes here too (exact fp32)
    x, w = rnd('scx2', (1, 32, 8, 32)), rnd('scw2', (2, 32, 3, 3), 0.1)
    y = ops.conv(dev(nhwc(x)), pack(w), None, pro_act=L.PRO_SWISH, act=L.ACT_SIGMOID, mma=L.MMA_BF16)
    check(nchw(y), torch.sigmoid(F.conv2d(x * torch.sigmoid(x), w, None, padding=1)), what='cout2 swish/sigmoid')


def test_conv3x3_splitk_matches():
    x, w, b = rnd('sx', (1, 512, 16, 16)), rnd('sw', (512, 512, 3, 3), 0.02), rnd('sb', (512,))
    ref = F.conv2d(x, w, b, padding=1)
    for sk in (1, 4, 9, 32):
        check(nchw(ops.conv(dev(nhwc(x)), pack(w), dev(b), split_k=sk)), ref, what=f'split_k={sk}')
    check(nchw(ops.conv(dev(nhwc(x)), pack(w), dev(b))), ref, what='auto split')


def test_conv_prologue_groupnorm_swish_residual():
    """the ResBlock pattern: GN(32, eps 1e-6) + swish folded into the conv prologue, residual in the epilogue."""
    x = rnd('gx', (2, 64, 32, 32), 2.0) + 0.5
    gamma, beta = rnd('gg', (64,)) * 0.2 + 1, rnd('gb', (64,)) * 0.2
    w, b, res = rnd('gw', (128, 64, 3, 3), 0.05), rnd('gbi', (128,)), rnd('gr', (2, 128, 32, 32))
    xd = dev(nhwc(x))
    pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
    y = ops.conv(xd, pack(w), dev(b), pro=pro, pro_act=L.PRO_SWISH, residual=dev(nhwc(res)))
    h = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    ref = F.conv2d(h * torch.sigmoid(h), w, b, padding=1) + res
    check(nchw(y), ref, what='gn+swish+conv+res')


def test_conv_instance_norm_relu_prologue():
    x = rnd('ix', (3, 96, 16, 16), 3.0) - 1.0
    w = rnd('iw', (96, 96, 3, 3), 0.05)
    xd = dev(nhwc(x))
    pro = ops.norm_affine(xd, None, None, 96, 1e-5)
    y = ops.conv(xd, pack(w), None, pro=pro, pro_act=L.PRO_RELU)
    check(nchw(y), F.conv2d(F.relu(F.instance_norm(x, eps=1e-5)), w, None, padding=1), what='in+relu+conv')


def test_conv_downsample_and_upsample():
    x, w, b = rnd('dx', (2, 128, 16, 16)), rnd('dw', (128, 128, 3, 3), 0.05), rnd('db', (128,))
    y = ops.conv(dev(nhwc(x)), pack(w), dev(b), down=True)
    check(nchw(y), F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2), what='downsample')
    y = ops.conv(dev(nhwc(x)), pack(w), dev(b), upsample=True)
    check(nchw(y), F.conv2d(F.interpolate(x, scale_factor=2.0, mode='nearest'), w, b, padding=1), what='upsample')


def test_conv_7x7_s2_cin3_and_cin130_and_small_cout():
    x, w = rnd('7x', (2, 3, 64, 64)), rnd('7w', (64, 3, 7, 7), 0.1)
    check(nchw(ops.conv(dev(nhwc(x)), pack(w), None, stride=2, pad=3, ksize=7)), F.conv2d(x, w, None, stride=2, padding=3),
          what='7x7s2')
    x, w, b = rnd('ux', (1, 130, 8, 8)), rnd('uw', (256, 130, 3, 3), 0.05), rnd('ub', (256,))
    check(nchw(ops.conv(dev(nhwc(x)), pack(w), dev(b), act=L.ACT_RELU)), F.relu(F.conv2d(x, w, b, padding=1)), what='cin130')
    x, w, b = rnd('ox', (1, 64, 32, 32)), rnd('ow', (3, 64, 3, 3), 0.05), rnd('ob', (3,))
    check(nchw(ops.conv(dev(nhwc(x)), pack(w), dev(b))), F.conv2d(x, w, b, padding=1), what='cout3')
    x, w = rnd('3x', (2, 3, 32, 32)), rnd('3w', (64, 3, 3, 3), 0.2)
    check(nchw(ops.conv(dev(nhwc(x)), pack(w), None)), F.conv2d(x, w, None, padding=1), what='cin3')


def test_linear_epilogues_and_slices():
    x, w, b = rnd('lx', (300, 256)), rnd('lw', (512, 256), 0.06), rnd('lb', (512,))
    res = rnd('lr', (300, 512))
    check(ops.linear(dev(x), dev(w), dev(b), act=L.ACT_GELU), F.gelu(F.linear(x, w, b)), what='gelu')
    check(ops.linear(dev(x), dev(w), dev(b), residual=dev(res)), F.linear(x, w, b) + res, what='residual')
    check(ops.linear(dev(x), dev(w[:1, :]), dev(b[:1]), act=L.ACT_SIGMOID), torch.sigmoid(F.linear(x, w[:1], b[:1])), what='sigmoid')
    # channel-slice input: second half of a [M,512] buffer
    xx = rnd('lxx', (300, 512))
    y = ops.conv(dev(xx).view(1, 300, 1, 512), dev(w), dev(b), pad=0, ksize=1, cin=256, in_off=256)
    check(y.view(300, 512), F.linear(xx[:, 256:], w, b), what='in_off slice')


def test_conv_cft_epilogue():
    dec, aux = rnd('fd', (1, 64, 16, 16)), rnd('fa', (1, 64, 16, 16))
    x, w, b = rnd('fx', (1, 64, 16, 16)), rnd('fw', (64, 64, 3, 3), 0.05), rnd('fb', (64,))
    y = ops.conv(dev(nhwc(x)), pack(w), dev(b), residual=dev(nhwc(dec)), aux=dev(nhwc(aux)), aux_w=0.7)
    check(nchw(y), dec + 0.7 * (dec * aux + F.conv2d(x, w, b, padding=1)), what='cft epilogue')
    y = ops.conv(dev(nhwc(x)), pack(w), dev(b), act=L.ACT_LRELU02)
    check(nchw(y), F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2), what='lrelu')


# ------------------------------------------------------------------------------------------------ attention
def ref_attn(q, k, v, scale, mask=None):
    s = torch.einsum('bhqd,bhkd->bhqk', q, k) * scale
    if mask is not None:
        s = s + mask
    return torch.einsum('bhqk,bhkd->bhqd', s.softmax(-1), v)


@pytest.mark.parametrize("B,H,Lq,Lk,D,Dv", [(2, 8, 256, 256, 64, 64), (1, 1, 64, 64, 512, 512), (2, 4, 96, 96, 256, 256),
                                            (3, 2, 50, 77, 48, 48), (1, 1, 256, 256, 128, 2), (2, 1, 40, 40, 128, 128)])
def test_attention_plain(B, H, Lq, Lk, D, Dv):
    q, k, v = rnd('aq', (B, Lq, H, D)), rnd('ak', (B, Lk, H, D)), rnd('av', (B, Lk, H, Dv))
    scale = D ** -0.5 * 3.0
    o = torch.empty(B, Lq, H, Dv, device='cuda')
    ops.attention(dev(q), dev(k), dev(v), o, B=B, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=scale,
                  q_str=(Lq * H * D, H * D, D), k_str=(Lk * H * D, H * D, D), v_str=(Lk * H * Dv, H * Dv, Dv),
                  o_str=(Lq * H * Dv, H * Dv, Dv))
    ref = ref_attn(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), scale).permute(0, 2, 1, 3)
    check(o, ref, what=f'attn {B, H, Lq, Lk, D, Dv}')


def test_attention_packed_qkv_and_peaked_softmax():
    """q|k|v packed in one buffer (strided heads) and large-magnitude scores (online-softmax rescale path)."""
    B, Lt, H, D = 2, 160, 4, 64
    qkv = rnd('pq', (B, Lt, 3 * H * D), 4.0)
    o = torch.empty(B, Lt, H * D, device='cuda')
    qd = dev(qkv)
    s3 = (Lt * 3 * H * D, 3 * H * D, D)
    ops.attention(qd, ops.offset(qd, H * D), ops.offset(qd, 2 * H * D), o, B=B, H=H, Lq=Lt, Lk=Lt, D=D, Dv=D, scale=1.0,
                  q_str=s3, k_str=s3, v_str=s3, o_str=(Lt * H * D, H * D, D))
    q, k, v = (t.reshape(B, Lt, H, D).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    check(o.view(B, Lt, H, D), ref_attn(q, k, v, 1.0).permute(0, 2, 1, 3), what='packed/peaked')


def test_attention_sparse_causal_mode():
    """KA:704-716: keys of frame f = [tokens of frame 0 ; tokens of frame max(f-1,0)]."""
    Bc, T, Lt, H, D = 2, 3, 64, 8, 48
    inner = H * D
    qkv = rnd('sq', (Bc * T, Lt, 3 * inner))
    qd = dev(qkv)
    o = torch.empty(Bc * T, Lt, inner, device='cuda')
    s3 = (Lt * 3 * inner, 3 * inner, D)
    ops.attention(qd, ops.offset(qd, inner), ops.offset(qd, 2 * inner), o, B=Bc * T, H=H, Lq=Lt, Lk=2 * Lt, D=D, Dv=D,
                  scale=D ** -0.5, q_str=s3, k_str=s3, v_str=s3, o_str=(Lt * inner, inner, D), mode=1, T=T, seg_len=Lt)
    q, k, v = qkv.chunk(3, dim=-1)
    former = torch.arange(T) - 1
    former[0] = 0

    def gather(t):
        t = t.reshape(Bc, T, Lt, inner)
        return torch.cat([t[:, [0] * T], t[:, former]], dim=2).reshape(Bc * T, 2 * Lt, inner)

    hs = lambda t: t.reshape(t.shape[0], t.shape[1], H, D).permute(0, 2, 1, 3)  # noqa: E731
    ref = ref_attn(hs(q), hs(gather(k)), hs(gather(v)), D ** -0.5).permute(0, 2, 1, 3).reshape(Bc * T, Lt, inner)
    check(o, ref, what='sparse causal')


@pytest.mark.parametrize("T", [2, 3, 20])
def test_attention_temporal_strided(T):
    """KA:671-680: batch = spatial token, tokens = frames, read in place from [(f d) c]."""
    Lt, H, D = 16, 8, 48
    inner = H * D
    qkv = rnd('tq', (T, Lt, 3 * inner))
    qd = dev(qkv)
    o = torch.empty(T, Lt, inner, device='cuda')
    st = (3 * inner, Lt * 3 * inner, D)
    ops.attention(qd, ops.offset(qd, inner), ops.offset(qd, 2 * inner), o, B=Lt, H=H, Lq=T, Lk=T, D=D, Dv=D,
                  scale=D ** -0.5, q_str=st, k_str=st, v_str=st, o_str=(inner, Lt * inner, D))
    q, k, v = (t.permute(1, 0, 2).reshape(Lt, T, H, D).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    ref = ref_attn(q, k, v, D ** -0.5).permute(0, 2, 1, 3).reshape(Lt, T, inner).permute(1, 0, 2)
    check(o, ref, what=f'temporal T={T}')


@pytest.mark.parametrize("shift", [0, 1])
def test_attention_swin_windows(shift):
    """GM/transformer.py:46-105 incl. roll, 2x2 windows, -100 region mask and the [f0;f1]/[f1;f0] key swap."""
    P, h8, w8, C = 2, 8, 8, 128
    n_img, Lt = 2 * P, h8 * w8
    q, k, v = rnd('wq', (n_img, Lt, C)), rnd('wk', (n_img, Lt, C)), rnd('wv', (n_img, Lt, C))
    sh = (h8 // 2) // 2 if shift else 0
    o = torch.empty(n_img, Lt, C, device='cuda')
    s = (Lt * C, C, 0)
    ops.attention(dev(q), dev(k), dev(v), o, B=n_img * 4, H=1, Lq=Lt // 4, Lk=Lt // 4, D=C, Dv=C, scale=1 / C ** 0.5,
                  q_str=s, k_str=s, v_str=s, o_str=s, mode=2, img_h=h8, img_w=w8, ksplit=2, shift=sh, kv_rot=P, n_img=n_img)
    mask = O.shift_window_mask(h8, w8, h8 // 2, w8 // 2, h8 // 4, w8 // 4)
    kr, vr = torch.cat([k[P:], k[:P]]), torch.cat([v[P:], v[:P]])
    ref = O._window_attention(q, kr, vr, 2, bool(shift), h8, w8, mask)
    check(o, ref, what=f'swin shift={sh}')


# ------------------------------------------------------------------------------------------------ small kernels
def test_layernorm_variants():
    x, g, b = rnd('nx', (300, 512), 3.0) + 1.0, rnd('ng', (512,)) * 0.2 + 1, rnd('nb', (512,)) * 0.2
    res, pos = rnd('nr', (300, 512)), rnd('np', (100, 512))
    ref = F.layer_norm(x, (512,), g, b, 1e-5)
    check(ops.layernorm(dev(x), dev(g), dev(b)), ref, 2e-5, 'ln')
    check(ops.layernorm(dev(x), dev(g), dev(b), res=dev(res)), ref + res, 2e-5, 'ln+res')
    y, y2 = ops.layernorm(dev(x), dev(g), dev(b), pos=dev(pos))
    check(y, ref, 2e-5, 'ln (dual)')
    check(y2, ref + pos.repeat(3, 1), 2e-5, 'ln+pos')
    x = rnd('nx2', (70, 128))
    check(ops.layernorm(dev(x), dev(g[:128]), dev(b[:128])), F.layer_norm(x, (128,), g[:128], b[:128], 1e-5), 2e-5, 'ln128')


def test_geglu_concat_addbcast():
    x = rnd('gx', (77, 2048), 2.0)
    h, g = x.chunk(2, dim=-1)
    check(ops.geglu(dev(x)), h * F.gelu(g), 1e-5, 'geglu')
    a, b = rnd('ca', (50, 2)), rnd('cb', (50, 128))
    assert torch.equal(ops.concat2(dev(a), dev(b)).cpu(), torch.cat([a, b], -1))
    t = rnd('bt', (64, 128))
    assert torch.allclose(ops.add_bcast(dev(rnd('ba', (3, 64, 128))), dev(t), -1.0).cpu(), rnd('ba', (3, 64, 128)) - t)


def test_argmax_gather_and_ties():
    logits = rnd('al', (512, 1024), 4.0)
    logits[7, 100] = logits[7, 900] = 50.0            # exact tie -> lowest index
    cb = rnd('acb', (1024, 256))
    out = torch.empty(512, 256, device='cuda')
    idx = torch.empty(512, dtype=torch.int32, device='cuda')
    margin = torch.empty(512, device='cuda')
    status = torch.zeros(1, dtype=torch.int32, device='cuda')
    L.call('keep_argmax_gather', dev(logits), dev(cb), None, idx, margin, out, 512, 1024, 256, status)
    ref_idx = logits.argmax(-1)
    ref_idx[7] = 100
    assert torch.equal(idx.cpu().long(), ref_idx)
    assert torch.equal(out.cpu(), cb[ref_idx])
    assert int(status.item()) == 0
    top2 = logits.topk(2, -1).values
    assert torch.allclose(margin.cpu(), top2[:, 0] - top2[:, 1], atol=1e-6)
    force = torch.arange(512, dtype=torch.int32, device='cuda') % 1024
    L.call('keep_argmax_gather', dev(logits), dev(cb), force, idx, None, out, 512, 1024, 256, None)
    assert torch.equal(out.cpu(), cb[force.cpu().long()])


def test_argmax_never_launders_non_finite_logits():
    """A NaN / inf logit row (an fp16-range overflow upstream under the x3 policy) must not become a plausible code: rows
    that are all NaN, rows with ONE NaN among finite logits, and rows whose maximum is +inf raise the status word and get
    a NaN-filled codebook row; the finite rows of the same launch are untouched."""
    logits = rnd('nl', (64, 1024), 4.0)
    logits[3] = float('nan')                       # all NaN: round 2 returned code 0 and a finite row
    logits[10, 517] = float('nan')                 # one NaN: the scan would just skip it
    logits[20, 5] = float('inf')                   # an overflow that stayed inf
    logits[30] = float('-inf')
    cb = rnd('ncb', (1024, 256))
    out = torch.empty(64, 256, device='cuda')
    idx = torch.empty(64, dtype=torch.int32, device='cuda')
    status = torch.zeros(1, dtype=torch.int32, device='cuda')
    L.call('keep_argmax_gather', dev(logits), dev(cb), None, idx, None, out, 64, 1024, 256, status)
    assert int(status.item()) & L.STATUS_NONFINITE_LOGITS
    o = out.cpu()
    bad = [3, 10, 20, 30]
    good = [r for r in range(64) if r not in bad]
    assert torch.isnan(o[bad]).all()
    assert torch.equal(o[good], cb[logits[good].argmax(-1)])
    assert ((idx.cpu() >= 0) & (idx.cpu() < 1024)).all()
    # clean launch: the word stays 0; keep_nonfinite_flag raises its own bit on a tensor with one inf / one NaN at any position
    status.zero_()
    L.call('keep_argmax_gather', dev(logits[good]), dev(cb), None, idx, None, out, len(good), 1024, 256, status)
    assert int(status.item()) == 0
    for n, pos, val in [(1 << 20, 12345, float('inf')), (4099, 4098, float('nan')), (7, 6, float('-inf')), (1 << 20, None, 0.0)]:
        t = torch.ones(n, device='cuda')
        if pos is not None:
            t[pos] = val
        status.zero_()
        L.call('keep_nonfinite_flag', t, n, status)
        assert int(status.item()) == (L.STATUS_NONFINITE_TENSOR if pos is not None else 0), (n, pos, val)


def test_relu_and_flow_warp_keep_nan():
    """The other two places a NaN could have vanished: ReLU written as v > 0 ? v : 0 (NaN -> 0) and grid_sample's 'outside
    the image' branch (NaN flow -> zeros).  Both propagate now (torch's F.relu and F.grid_sample do as well)."""
    x = rnd('rn', (1, 64, 16, 16))
    x[0, 5, 3, 3] = float('nan')
    w = rnd('rnw', (64, 64, 1, 1), 0.1)
    y = ops.conv(dev(nhwc(x)), pack(w), None, pad=0, ksize=1, act=L.ACT_RELU)
    assert torch.isnan(nchw(y)[0, :, 3, 3]).all() and torch.isfinite(nchw(y)[0, :, 4, 4]).all()
    sc, sh = torch.ones(1, 64, device='cuda'), torch.zeros(1, 64, device='cuda')
    y = ops.conv(dev(nhwc(x)), pack(rnd('rnw3', (64, 64, 3, 3), 0.05)), None, pro=(sc, sh), pro_act=L.PRO_RELU)
    assert torch.isnan(nchw(y)[0, :, 3, 3]).all()
    img = dev(nhwc(rnd('wi', (1, 3, 32, 32))))
    flow = torch.zeros(1, 32, 32, 2, device='cuda')
    flow[0, 7, 9, 0] = float('nan')
    flow[0, 8, 9, 1] = float('inf')
    out = torch.empty_like(img)
    L.call('keep_flow_warp', img, flow, out, 1, 32, 32, 3)
    assert torch.isnan(out[0, 7, 9]).all() and torch.isnan(out[0, 8, 9]).all()
    keep = torch.ones(32, 32, dtype=torch.bool)
    keep[7, 9] = keep[8, 9] = False
    assert torch.allclose(out[0].cpu()[keep], img[0].cpu()[keep], atol=1e-5)      # zero flow elsewhere: the image itself


def test_vq_nearest(synth_weights):
    z = op_input('vq_nn', (1, 256, 8, 8), 0.7)
    cb = synth_weights['quantize.embedding.weight']
    idx = torch.empty(64, dtype=torch.int32, device='cuda')
    L.call('keep_vq_nearest', dev(nhwc(z)).view(64, 256), dev(cb), idx, 64, 1024, 256)
    gold = np.load(os.path.join(GOLDEN, 'ops.npz'))['vq_nn_idx']
    assert np.array_equal(idx.cpu().numpy(), gold)


def test_vq_nearest_at_scale_ragged_and_ties(synth_weights):
    """8f-3 at the size of a bench step (M = 256 tokens x 16 clips x 20 frames), ragged M / codebook sizes, exact ties."""
    cb = dev(synth_weights['quantize.embedding.weight'])                         # [1024, 256]
    M = 256 * 16 * 20 + 37                                                       # not a multiple of the 64-token tile
    z = (cb[torch.randint(0, 1024, (M,), device='cuda', generator=torch.Generator('cuda').manual_seed(3))] +
         0.02 * torch.randn(M, 256, device='cuda', generator=torch.Generator('cuda').manual_seed(4)))
    idx = torch.empty(M, dtype=torch.int32, device='cuda')
    L.call('keep_vq_nearest', z, cb, idx, M, 1024, 256)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        L.call('keep_vq_nearest', z, cb, idx, M, 1024, 256)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 5
    print(f'keep_vq_nearest M={M}: {ms * 1e3:.0f} us, {2.0 * M * 1024 * 256 / ms / 1e9:.1f} TFLOP/s (exact-f32 MFMA), '
          f'{(M * 256 * 4 + M * 4) / ms / 1e6:.1f} GB/s of token traffic')
    d = ((z.double() ** 2).sum(1, keepdim=True) + (cb.double() ** 2).sum(1)) - 2 * z.double() @ cb.double().t()   # VQ:43-44
    top2 = torch.topk(d, 2, dim=1, largest=False)
    margin = top2.values[:, 1] - top2.values[:, 0]
    ref = top2.indices[:, 0].to(torch.int32)
    agree = idx == ref
    assert agree[margin > 1e-5 * d.abs().max()].all() and agree.float().mean() > 0.9999, float(agree.float().mean())
    # ragged codebook (1000 codes: the last 128-code tile is partial) and a short dim
    cb2 = cb[:1000, :192].contiguous()
    z2 = z[:777, :192].contiguous()
    idx2 = torch.empty(777, dtype=torch.int32, device='cuda')
    L.call('keep_vq_nearest', z2, cb2, idx2, 777, 1000, 192)
    d2 = ((z2.double() ** 2).sum(1, keepdim=True) + (cb2.double() ** 2).sum(1)) - 2 * z2.double() @ cb2.double().t()
    assert (idx2 == d2.argmin(1).to(torch.int32)).float().mean() > 0.999
    # exact ties: duplicated code rows -> the lowest index wins (torch.argmin)
    cb3 = torch.cat([cb[:200], cb[:200], cb[200:400]]).contiguous()              # rows j and j + 200 identical for j < 200
    z3 = cb3[torch.arange(200, 400, device='cuda')].contiguous()                 # each token equals its code exactly
    idx3 = torch.empty(200, dtype=torch.int32, device='cuda')
    L.call('keep_vq_nearest', z3, cb3, idx3, 200, 600, 256)
    assert torch.equal(idx3.cpu(), torch.arange(0, 200, dtype=torch.int32))


def test_kalman_update_and_flow_warp():
    zc, zp, g = rnd('ku_z', (1, 256, 8, 8)), rnd('ku_zp', (1, 256, 8, 8)), (rnd('ku_g', (1, 1, 8, 8)) + 1) / 2
    out = torch.empty(1, 8, 8, 256, device='cuda')
    L.call('keep_kalman_update', dev(nhwc(zc)), dev(nhwc(zp)), dev(g.view(1, 64)), out, 1, 64, 256)
    check(nchw(out), torch.from_numpy(np.load(os.path.join(GOLDEN, 'ops.npz'))['kalman_update']), 1e-6, 'kalman update')
    img, flo = rnd('warp_img', (2, 3, 32, 32)), rnd('warp_flow', (2, 32, 32, 2), 6.0)
    out = torch.empty(2, 32, 32, 3, device='cuda')
    L.call('keep_flow_warp', dev(nhwc(img)), dev(flo), out, 2, 32, 32, 3)
    check(nchw(out), torch.from_numpy(np.load(os.path.join(GOLDEN, 'ops.npz'))['warp']), 1e-5, 'flow warp (golden)')
    big = rnd('warp_flow2', (2, 32, 32, 2), 40.0)      # mostly out of range -> zero padding
    L.call('keep_flow_warp', dev(nhwc(img)), dev(big), out, 2, 32, 32, 3)
    check(nchw(out), O.flow_warp(img, big), 1e-5, 'flow warp (far)')


def test_convex_upsample_and_layouts():
    mask, flow = rnd('um', (2, 576, 6, 5), 3.0), rnd('uf', (2, 2, 6, 5), 5.0)
    out = torch.empty(2, 48, 40, 2, device='cuda')
    L.call('keep_convex_upsample', dev(nhwc(mask)), dev(nhwc(flow)), out, 2, 6, 5, 8)
    m = torch.softmax(mask.view(2, 1, 9, 8, 8, 6, 5), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(2, 2, 9, 1, 1, 6, 5)
    ref = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(2, 2, 48, 40)
    check(nchw(out), ref, 1e-5, 'convex upsample')
    x = rnd('lx', (3, 3, 40, 24))
    assert torch.equal(ops.nchw_to_nhwc(dev(x)).cpu(), nhwc(x))
    assert torch.equal(ops.nhwc_to_nchw(dev(nhwc(x))).cpu(), x)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    check(nchw(ops.nchw_to_nhwc(dev(x), mode=1)), (((x + 1) / 2 * 255) / 255. - mean) / std, 1e-6, 'gmflow prep')


def test_tensor2img_img2tensor_bit_exact():
    """P4 on the device (keep_img2tensor / keep_tensor2img) against the ORACLE (oracle/converters_oracle.py) and the golden
    outputs of the reference's own img2tensor / tensor2img (tests/golden/ops.npz: conv_in_crops, conv_out_u8), bit for bit."""
    import converters_oracle as CO
    g = np.load(os.path.join(GOLDEN, 'ops.npz'))
    x = rnd('t2i', (2, 3, 64, 64), 1.3)
    x[0, :, 0, :8] = torch.tensor([-1.2, -1.0, -0.5 / 255, 0.0, 1.0 / 255, 1.0, 1.3, 0.00392156862])
    out = torch.empty(2, 64, 64, 3, dtype=torch.uint8, device='cuda')
    L.call('keep_tensor2img', dev(nhwc(x)), out, 2 * 64 * 64)
    for n in range(2):
        assert np.array_equal(out[n].cpu().numpy(), CO.net_output_to_bgr_u8(x[n].numpy()))
        assert np.array_equal(out[n].cpu().numpy(), g['conv_out_u8'][n])
    # a dense sweep around every rounding boundary of the uint8 map: k / 255 steps +- a few float32 ulps, half-way points
    k = torch.arange(0, 256, dtype=torch.float64)
    pts = torch.cat([(k / 255.0 * 2 - 1), ((k + 0.5) / 255.0 * 2 - 1)]).float()
    sweep = torch.cat([pts, torch.nextafter(pts, torch.tensor(2.0)), torch.nextafter(pts, torch.tensor(-2.0))])
    sweep = sweep[: (sweep.numel() // 3) * 3].view(1, 3, 1, -1).contiguous()
    o2 = torch.empty(1, 1, sweep.shape[-1], 3, dtype=torch.uint8, device='cuda')
    L.call('keep_tensor2img', dev(nhwc(sweep)), o2, sweep.shape[-1])
    assert np.array_equal(o2[0].cpu().numpy(), CO.net_output_to_bgr_u8(sweep[0].numpy()))
    crops = [synth.ramp_image(64, 64), synth.ramp_image(64, 64)[::-1].copy()]
    u8 = torch.from_numpy(np.stack(crops)).cuda()
    f = torch.empty(2, 64, 64, 3, device='cuda')
    L.call('keep_img2tensor', u8, f, 2 * 64 * 64)
    assert np.array_equal(nchw(f).cpu().numpy(), CO.crops_to_net_input(crops))
    assert np.array_equal(nchw(f).cpu().numpy(), g['conv_in_crops'])
    allv = np.arange(256, dtype=np.uint8).repeat(3).reshape(1, 16, 16, 3)          # every uint8 value
    fa = torch.empty(1, 16, 16, 3, device='cuda')
    L.call('keep_img2tensor', torch.from_numpy(allv).cuda(), fa, 256)
    assert np.array_equal(nchw(fa).cpu().numpy(), CO.crops_to_net_input(list(allv)))


def test_bad_arguments_fail_loudly():
    with pytest.raises(L.KeepHipError, match='dtype'):
        x = torch.zeros(1, 8, 8, 16, device='cuda')
        L.conv2d(inp=x, weight=x, out=x, N=1, H=8, W=8, Cin=16, Cout=16, KH=1, KW=1, stride=1, Ho=8, Wo=8, in_ld=16,
                 out_ld=16, split_k=1, dtype=L.BF16)
    with pytest.raises(L.KeepHipError, match='even'):
        q = torch.zeros(4, 33, device='cuda')
        ops.attention(q, q, q, q, B=1, H=1, Lq=4, Lk=4, D=33, Dv=33, scale=1.0, q_str=(0, 33, 0), k_str=(0, 33, 0),
                      v_str=(0, 33, 0), o_str=(0, 33, 0))


# ------------------------------------------------------------------------------------------------ bf16 MFMA policy
def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("cin,cout,hw,n,k", [(64, 64, 32, 2, 3), (128, 256, 16, 1, 3), (96, 128, 16, 2, 3), (96, 96, 32, 2, 3),
                                             (256, 128, 64, 1, 3), (512, 512, 16, 1, 1), (130, 256, 8, 1, 3),
                                             (3, 64, 32, 2, 3), (64, 3, 24, 1, 3)])
def test_conv_bf16_operands_exact_products(cin, cout, hw, n, k):
    """bf16 policy: operands are RNE-rounded to bf16 when staged, products accumulate in fp32 -> equals an fp32 conv of
    the rounded operands up to accumulation order."""
    x, w, b = rnd('bx', (n, cin, hw, hw)), rnd('bw', (cout, cin, k, k), 0.05), rnd('bb', (cout,))
    wp = pack(w)
    y = ops.conv(dev(nhwc(x)), wp, dev(b), pad=k // 2, ksize=k, mma=L.MMA_BF16, wb=wp.to(torch.bfloat16))
    check(nchw(y), F.conv2d(bf16r(x), bf16r(w), b, padding=k // 2), 2e-5, f'bf16 conv {cin}->{cout} k{k}')


def test_conv_bf16_prologue_epilogue_splitk_upsample():
    x = rnd('gx', (2, 64, 32, 32), 2.0) + 0.5
    gamma, beta = rnd('gg', (64,)) * 0.2 + 1, rnd('gb', (64,)) * 0.2
    w, b, res = rnd('gw', (128, 64, 3, 3), 0.05), rnd('gbi', (128,)), rnd('gr', (2, 128, 32, 32))
    xd, wp = dev(nhwc(x)), pack(w)
    wb = wp.to(torch.bfloat16)
    pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
    h = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    ref = F.conv2d(bf16r(h * torch.sigmoid(h)), bf16r(w), b, padding=1) + res
    for sk in (1, 3):
        y = ops.conv(xd, wp, dev(b), pro=pro, pro_act=L.PRO_SWISH, residual=dev(nhwc(res)), mma=L.MMA_BF16, wb=wb, split_k=sk)
        # the activated value is rounded to bf16 after a fast-exp swish: a 1-ulp input difference moves a bf16 rounding
        check(nchw(y), ref, 2e-3, f'bf16 gn+swish+conv+res split_k={sk}')
    y = ops.conv(xd, wp, dev(b), upsample=True, mma=L.MMA_BF16, wb=wb)
    check(nchw(y), F.conv2d(bf16r(F.interpolate(x, scale_factor=2.0, mode='nearest')), bf16r(w), b, padding=1), 2e-5, 'bf16 up')
    y = ops.conv(xd, wp, dev(b), down=True, mma=L.MMA_BF16, wb=wb)
    check(nchw(y), F.conv2d(F.pad(bf16r(x), (0, 1, 0, 1)), bf16r(w), b, stride=2), 2e-5, 'bf16 down')


def test_conv_bf16_halo_paths():
    """3x3 halo kernel: fp32 input, bf16 pre-activated input (GN + swish via keep_norm_act_bf16), upsampled input,
    channel-slice input, CFT epilogue, split-K -- against fp32 convs of the bf16-rounded operands."""
    x = rnd('hx', (2, 64, 32, 64), 2.0) + 0.3
    w, b = rnd('hw', (128, 64, 3, 3), 0.05), rnd('hb', (128,))
    xd, wp = dev(nhwc(x)), pack(w)
    wb = wp.to(torch.bfloat16)
    for sk in (1, 2):
        y = ops.conv(xd, wp, dev(b), mma=L.MMA_BF16, wb=wb, split_k=sk)
        check(nchw(y), F.conv2d(bf16r(x), bf16r(w), b, padding=1), 2e-5, f'halo fp32-in split_k={sk}')
    gamma, beta = rnd('hg', (64,)) * 0.2 + 1, rnd('hbt', (64,)) * 0.2
    pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
    res = rnd('hr', (2, 128, 32, 64))
    y = ops.conv(xd, wp, dev(b), pro=pro, pro_act=L.PRO_SWISH, residual=dev(nhwc(res)), mma=L.MMA_BF16, wb=wb)
    h = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    check(nchw(y), F.conv2d(bf16r(h * torch.sigmoid(h)), bf16r(w), b, padding=1) + res, 2e-3, 'halo gn+swish')
    xs = rnd('hxs', (1, 64, 16, 16))
    y = ops.conv(dev(nhwc(xs)), wp, dev(b), upsample=True, mma=L.MMA_BF16, wb=wb)
    check(nchw(y), F.conv2d(bf16r(F.interpolate(xs, scale_factor=2.0, mode='nearest')), bf16r(w), b, padding=1), 2e-5,
          'halo upsample')
    xw = rnd('hxw', (1, 128, 32, 32))            # slice: second 64 channels of a 128-wide buffer
    dec, aux = rnd('hd', (1, 128, 32, 32)), rnd('ha', (1, 128, 32, 32))
    y = ops.conv(dev(nhwc(xw)), wp, dev(b), cin=64, in_off=64, residual=dev(nhwc(dec)), aux=dev(nhwc(aux)), aux_w=1.0,
                 mma=L.MMA_BF16, wb=wb)
    check(nchw(y), dec + dec * aux + F.conv2d(bf16r(xw[:, 64:]), bf16r(w), b, padding=1), 2e-5, 'halo slice + cft')


def test_group_stats_small_matches_two_pass():
    x = rnd('gsx', (3, 512, 16, 16), 2.0) + 0.7
    gamma, beta = rnd('gsg', (512,)) * 0.2 + 1, rnd('gsb', (512,)) * 0.2
    xd = dev(nhwc(x))
    sc, sh = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)       # one-launch path (small map)
    h = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    got = nhwc(x) * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]
    check(got, nhwc(h), 2e-5, 'group stats small')
    sc2, sh2 = ops.norm_affine(xd, None, None, 512, 1e-5)                # instance norm, C groups
    got = nhwc(x) * sc2.cpu()[:, None, None, :] + sh2.cpu()[:, None, None, :]
    check(got, nhwc(F.instance_norm(x, eps=1e-5)), 2e-5, 'instance stats small')


@pytest.mark.parametrize("B,H,Lq,Lk,D,Dv", [(2, 8, 256, 256, 64, 64), (1, 1, 64, 64, 512, 512), (2, 4, 96, 96, 256, 256),
                                            (3, 2, 50, 77, 48, 48), (1, 1, 256, 256, 128, 2), (2, 1, 40, 40, 128, 128),
                                            (4, 8, 20, 20, 48, 48)])
def test_attention_bf16_operands(B, H, Lq, Lk, D, Dv):
    """bf16 policy: Q, K, V (and P) rounded to bf16, fp32 softmax/accumulate -> within bf16 resolution of the fp32
    attention of the rounded operands."""
    q, k, v = rnd('aq', (B, Lq, H, D)), rnd('ak', (B, Lk, H, D)), rnd('av', (B, Lk, H, Dv))
    scale = D ** -0.5 * 3.0
    o = torch.empty(B, Lq, H, Dv, device='cuda')
    ops.attention(dev(q), dev(k), dev(v), o, B=B, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=scale,
                  q_str=(Lq * H * D, H * D, D), k_str=(Lk * H * D, H * D, D), v_str=(Lk * H * Dv, H * Dv, Dv),
                  o_str=(Lq * H * Dv, H * Dv, Dv), mma=L.MMA_BF16)
    ref = ref_attn(bf16r(q).permute(0, 2, 1, 3), bf16r(k).permute(0, 2, 1, 3), bf16r(v).permute(0, 2, 1, 3), scale)
    check(o, ref.permute(0, 2, 1, 3), 6e-3, what=f'bf16 attn {B, H, Lq, Lk, D, Dv}')


def test_attention_bf16_window_and_sparse_modes():
    P, h8, w8, C = 2, 8, 8, 128
    n_img, Lt = 2 * P, h8 * w8
    q, k, v = rnd('wq', (n_img, Lt, C)), rnd('wk', (n_img, Lt, C)), rnd('wv', (n_img, Lt, C))
    o = torch.empty(n_img, Lt, C, device='cuda')
    s = (Lt * C, C, 0)
    ops.attention(dev(q), dev(k), dev(v), o, B=n_img * 4, H=1, Lq=Lt // 4, Lk=Lt // 4, D=C, Dv=C, scale=1 / C ** 0.5,
                  q_str=s, k_str=s, v_str=s, o_str=s, mode=2, img_h=h8, img_w=w8, ksplit=2, shift=2, kv_rot=P, n_img=n_img,
                  mma=L.MMA_BF16)
    mask = O.shift_window_mask(h8, w8, h8 // 2, w8 // 2, h8 // 4, w8 // 4)
    kr, vr = torch.cat([k[P:], k[:P]]), torch.cat([v[P:], v[:P]])
    check(o, O._window_attention(bf16r(q), bf16r(kr), bf16r(vr), 2, True, h8, w8, mask), 6e-3, 'bf16 swin')
    Bc, T, Lt, H, D = 2, 3, 64, 8, 48
    inner = H * D
    qkv = rnd('sq', (Bc * T, Lt, 3 * inner))
    qd = dev(qkv)
    o = torch.empty(Bc * T, Lt, inner, device='cuda')
    s3 = (Lt * 3 * inner, 3 * inner, D)
    ops.attention(qd, ops.offset(qd, inner), ops.offset(qd, 2 * inner), o, B=Bc * T, H=H, Lq=Lt, Lk=2 * Lt, D=D, Dv=D,
                  scale=D ** -0.5, q_str=s3, k_str=s3, v_str=s3, o_str=(Lt * inner, inner, D), mode=1, T=T, seg_len=Lt,
                  mma=L.MMA_BF16)
    qq, kk, vv = bf16r(qkv).chunk(3, dim=-1)
    former = torch.arange(T) - 1
    former[0] = 0

    def gather(t):
        t = t.reshape(Bc, T, Lt, inner)
        return torch.cat([t[:, [0] * T], t[:, former]], dim=2).reshape(Bc * T, 2 * Lt, inner)

    hs = lambda t: t.reshape(t.shape[0], t.shape[1], H, D).permute(0, 2, 1, 3)  # noqa: E731
    ref = ref_attn(hs(qq), hs(gather(kk)), hs(gather(vv)), D ** -0.5).permute(0, 2, 1, 3).reshape(Bc * T, Lt, inner)
    check(o, ref, 6e-3, 'bf16 sparse causal')


@pytest.mark.parametrize("mma", [0, 1])
def test_conv_epilogue_stats_match_standalone(mma):
    """(sum, sumsq) emitted by the conv epilogue == statistics of the conv output (GroupNorm of the next layer)."""
    x, w, b = rnd('esx', (2, 64, 32, 32)), rnd('esw', (128, 64, 3, 3), 0.05), rnd('esb', (128,))
    res = rnd('esr', (2, 128, 32, 32))
    gamma, beta = rnd('esg', (128,)) * 0.2 + 1, rnd('esbt', (128,)) * 0.2
    wp = pack(w)
    kw = dict(mma=mma, wb=wp.to(torch.bfloat16)) if mma else {}
    y, st = ops.conv(dev(nhwc(x)), wp, dev(b), residual=dev(nhwc(res)), stats=True, split_k=1, **kw)
    assert st is not None
    sc, sh = ops.norm_affine(y, dev(gamma), dev(beta), 32, 1e-6, stats=st)
    sc2, sh2 = ops.norm_affine(y, dev(gamma), dev(beta), 32, 1e-6)      # no fused stats -> standalone kernels
    check(sc, sc2, 1e-5, 'scale'); check(sh, sh2, 1e-5, 'shift')
    h = F.group_norm(nchw(y).cpu(), 32, gamma, beta, eps=1e-6)
    got = y.cpu() * sc.cpu()[:, None, None, :] + sh.cpu()[:, None, None, :]
    check(got, nhwc(h), 2e-5, 'fused stats -> group norm')
    # 1x1 / strided producers through the gather kernels
    y, st = ops.conv(dev(nhwc(x)), dev(rnd('esw1', (96, 64)) * 0.1).view(96, 1, 1, 64), None, stride=2, pad=0, ksize=1, stats=True,
                     split_k=1, **(dict(mma=1, wb=(dev(rnd('esw1', (96, 64)) * 0.1)).to(torch.bfloat16)) if mma else {}))
    assert st is not None
    sc, sh = ops.norm_affine(y, None, None, 96, 1e-5, stats=st)
    sc2, sh2 = ops.norm_affine(y, None, None, 96, 1e-5)
    check(sc, sc2, 1e-5, 'in scale'); check(sh, sh2, 1e-5, 'in shift')


def test_conv_bf16_halo_bf16_output_and_bf16_inputs():
    """bf16 policy storage: (a) the halo kernel writes a bf16 tensor + fp32-accurate GroupNorm partials (ResBlock conv1),
    (b) the normalise pass reads that bf16 tensor and feeds the second halo conv."""
    x, w, b = rnd('hbx', (2, 64, 32, 32)), rnd('hbw', (128, 64, 3, 3), 0.05), rnd('hbb', (128,))
    wp = pack(w)
    wb = wp.to(torch.bfloat16)
    xd = dev(nhwc(x))
    y32, st32 = ops.conv(xd, wp, dev(b), mma=L.MMA_BF16, wb=wb, stats=True, split_k=1)
    y16, st16 = ops.conv(xd, wp, dev(b), mma=L.MMA_BF16, wb=wb, stats=True, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and st16 is not None
    check(y16.float(), bf16r(y32.cpu()), 1e-6, 'halo bf16 output == RNE(fp32 output)')
    check(st16.part, st32.part, 1e-6, 'stats taken before rounding')
    # (b) GN + swish pass from the bf16 tensor, then the second halo conv
    gamma, beta = rnd('hbg', (128,)) * 0.2 + 1, rnd('hbbt', (128,)) * 0.2
    w2 = rnd('hbw2', (64, 128, 3, 3), 0.05)
    pro = ops.norm_affine(y16, dev(gamma), dev(beta), 32, 1e-6, stats=st16)
    z = ops.conv(y16, pack(w2), None, pro=pro, pro_act=L.PRO_SWISH, mma=L.MMA_BF16, wb=pack(w2).to(torch.bfloat16))
    sc, sh = pro[0].cpu(), pro[1].cpu()
    hn = y16.float().cpu() * sc[:, None, None, :] + sh[:, None, None, :]
    hn = bf16r(hn * torch.sigmoid(hn))
    check(nchw(z), F.conv2d(nchw(hn), bf16r(w2), None, padding=1), 3e-4, 'bf16 tensor -> norm pass -> halo conv')


@pytest.mark.parametrize("N,out_bf16,with_bias", [(128, False, False), (384, True, False), (256, True, True), (128, False, True)])
def test_token_linear_streaming_gemm(N, out_bf16, with_bias):
    """keep_token_linear (GMFlow projections): persistent blocks, W resident in LDS, ragged M, fp32 / bf16 output."""
    M = 128 * 5 + 37
    x, w = rnd('tlx', (M, 128)), rnd('tlw', (N, 128), 0.08)
    b = rnd('tlb', (N,)) if with_bias else None
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16 if out_bf16 else torch.float32)
    L.call('keep_token_linear', dev(x), dev(w).to(torch.bfloat16), None if b is None else dev(b), out, M, 128, N,
           L.BF16 if out_bf16 else L.F32)
    ref = F.linear(bf16r(x), bf16r(w), b)
    if out_bf16:
        check(out.float(), bf16r(ref), 1e-2, f'token linear N={N} bf16 out')
    else:
        check(out, ref, 2e-5, f'token linear N={N}')


def test_layernorm_c128_streaming_variant():
    """keep_layernorm, C = 128 and M >= 4096 (GMFlow token stream): the 16-lanes-per-row kernel vs torch, with and
    without the residual, ragged M."""
    M = 4096 + 37
    x, res = rnd('l1x', (M, 128), 3.0) + 0.7, rnd('l1r', (M, 128))
    g, b = rnd('l1g', (128,)) * 0.3 + 1, rnd('l1b', (128,)) * 0.3
    ref = F.layer_norm(x, (128,), g, b, eps=1e-5)
    check(ops.layernorm(dev(x), dev(g), dev(b)), ref, 2e-5, 'LN C=128')
    check(ops.layernorm(dev(x), dev(g), dev(b), res=dev(res)), ref + res, 2e-5, 'LN C=128 + res')


def test_gm_mlp_fused_matches_two_gemms():
    """keep_gm_mlp (GM/transformer.py:139-142,182 fused): vs torch on the bf16-rounded operands with the GELU output
    rounded to bf16 (what the second MFMA consumes), ragged M."""
    C, M = 128, 300
    a, b = rnd('ma', (M, C)), rnd('mb', (M, C))
    w0, w2 = rnd('mw0', (8 * C, 2 * C), 0.06), rnd('mw2', (C, 8 * C), 0.03)
    out = torch.empty(M, C, device='cuda')
    L.call('keep_gm_mlp', dev(a), dev(b), dev(w0).to(torch.bfloat16), dev(w2).to(torch.bfloat16), out, M, C)
    h = F.gelu(F.linear(bf16r(torch.cat([a, b], -1)), bf16r(w0)))
    ref = F.linear(bf16r(h), bf16r(w2))
    check(out, ref, 2e-3, 'fused GMFlow FFN')
    # and against the unfused engine path (fp32 intermediate): difference = one bf16 rounding of the hidden activations
    hm = ops.linear(ops.concat2(dev(a), dev(b)), dev(w0), act=L.ACT_GELU, )
    check(out, ops.linear(hm, dev(w2)), 2e-2, 'fused vs two fp32-policy GEMMs')


def test_conv_bf16_rgb_first_conv_kernel(monkeypatch):
    """VQ conv_in (3 -> 64, 3x3): the persistent im2col-in-LDS kernel vs torch on the rounded operands, vs the flat-K
    gather kernel, and its fused GroupNorm partials; also Cout = 96 (masked half cout-block) and Cin = 1."""
    for (n, cin, cout, h, wd) in [(2, 3, 64, 32, 64), (1, 3, 96, 8, 32), (3, 1, 32, 16, 32)]:
        x, w, b = rnd('c3x', (n, cin, h, wd)), rnd('c3w', (cout, cin, 3, 3), 0.2), rnd('c3b', (cout,))
        wp = pack(w)
        wb = wp.to(torch.bfloat16)
        ops.DEFAULT.profile = []
        y, st = ops.conv(dev(nhwc(x)), wp, dev(b), mma=L.MMA_BF16, wb=wb, stats=True)
        assert ops.DEFAULT.profile[-1][0] == 'conv3x3_c3_kernel' and st is not None
        ops.DEFAULT.profile = None
        ops.DEFAULT.flags = L.CONV_NO_C3
        y_g, _ = ops.conv(dev(nhwc(x)), wp, dev(b), mma=L.MMA_BF16, wb=wb, stats=True)
        ops.DEFAULT.flags = 0
        check(nchw(y), F.conv2d(bf16r(x), bf16r(w), b, padding=1), 2e-5, f'rgb conv {cin}->{cout}')
        check(y, y_g, 2e-5, 'rgb conv vs flat-K gather kernel')
        sc, sh = ops.norm_affine(y, None, None, cout, 1e-5, stats=st)
        sc2, sh2 = ops.norm_affine(y, None, None, cout, 1e-5)
        check(sc, sc2, 1e-5, 'rgb conv fused stats scale'); check(sh, sh2, 1e-5, 'rgb conv fused stats shift')


def test_conv_bf16_halo_fused_prologue_variant(monkeypatch):
    """The C-ABI also accepts the GroupNorm affine + swish directly on the bf16 halo path (conv3x3_halo_kernel, applied
    while the fp32 halo is staged) -- the engine prefers the separate normalise pass, so exercise the fused form here."""
    monkeypatch.setattr(ops, 'HALO_PRENORM_MINPIX', 1 << 40)
    x, w, b = rnd('fpx', (2, 64, 32, 32), 2.0) + 0.5, rnd('fpw', (128, 64, 3, 3), 0.05), rnd('fpb', (128,))
    gamma, beta = rnd('fpg', (64,)) * 0.2 + 1, rnd('fpbt', (64,)) * 0.2
    res = rnd('fpr', (2, 128, 32, 32))
    xd = dev(nhwc(x))
    wp = pack(w)
    pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
    y, st = ops.conv(xd, wp, dev(b), pro=pro, pro_act=L.PRO_SWISH, residual=dev(nhwc(res)), mma=L.MMA_BF16,
                     wb=wp.to(torch.bfloat16), stats=True, split_k=1)
    assert st is not None
    hn = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    hn = bf16r(hn * torch.sigmoid(hn))
    check(nchw(y), F.conv2d(hn, bf16r(w), b, padding=1) + res, 3e-4, 'halo with fused GN+swish prologue')
    sc, sh = ops.norm_affine(y, None, None, 128, 1e-5, stats=st)
    sc2, sh2 = ops.norm_affine(y, None, None, 128, 1e-5)
    check(sc, sc2, 1e-5, 'fused-prologue halo stats scale'); check(sh, sh2, 1e-5, 'fused-prologue halo stats shift')


def test_conv_bf16_flat_k_small_cin():
    x, w = rnd('7x', (2, 3, 64, 64)), rnd('7w', (64, 3, 7, 7), 0.1)
    wp = pack(w)
    y = ops.conv(dev(nhwc(x)), wp, None, stride=2, pad=3, ksize=7, mma=L.MMA_BF16, wb=wp.to(torch.bfloat16))
    check(nchw(y), F.conv2d(bf16r(x), bf16r(w), None, stride=2, padding=3), 2e-5, 'flat-K 7x7 s2')
    x, w, b = rnd('3x', (2, 3, 32, 32)), rnd('3w', (64, 3, 3, 3), 0.2), rnd('3b', (64,))
    wp = pack(w)
    y, _ = ops.conv(dev(nhwc(x)), wp, dev(b), mma=L.MMA_BF16, wb=wp.to(torch.bfloat16), stats=True)
    check(nchw(y), F.conv2d(bf16r(x), bf16r(w), b, padding=1), 2e-5, 'flat-K 3x3')


@pytest.mark.parametrize("B,H,Lq,Lk,D,Dv", [(2, 8, 256, 256, 64, 64), (1, 1, 256, 256, 512, 512), (2, 4, 200, 200, 256, 256),
                                            (3, 8, 50, 77, 48, 48), (2, 1, 1024, 1024, 128, 128), (4, 8, 20, 20, 48, 48),
                                            (2, 8, 256, 512, 48, 48), (1, 2, 130, 70, 32, 128)])
def test_attention_bf16_inputs(B, H, Lq, Lk, D, Dv):
    """q/k/v stored as bf16 (projection GEMMs write bf16): 64-key tiles, table-driven offsets, register prefetch."""
    q, k, v = rnd('aq', (B, Lq, H, D)), rnd('ak', (B, Lk, H, D)), rnd('av', (B, Lk, H, Dv))
    scale = D ** -0.5 * 3.0
    o = torch.empty(B, Lq, H, Dv, device='cuda')
    qb, kb, vb = (dev(t).to(torch.bfloat16) for t in (q, k, v))
    ops.attention(qb, kb, vb, o, B=B, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=scale,
                  q_str=(Lq * H * D, H * D, D), k_str=(Lk * H * D, H * D, D), v_str=(Lk * H * Dv, H * Dv, Dv),
                  o_str=(Lq * H * Dv, H * Dv, Dv))
    ref = ref_attn(bf16r(q).permute(0, 2, 1, 3), bf16r(k).permute(0, 2, 1, 3), bf16r(v).permute(0, 2, 1, 3), scale)
    check(o, ref.permute(0, 2, 1, 3), 6e-3, what=f'bf16-in attn {B, H, Lq, Lk, D, Dv}')


def test_attention_bf16_inputs_window_sparse_and_linear_bf16_out():
    P, h8, w8, C = 2, 16, 16, 128
    n_img, Lt = 2 * P, h8 * w8
    x = rnd('wx', (n_img * Lt, C))
    wqkv = rnd('wqkv', (3 * C, C), 0.1)
    qkv16 = ops.conv(dev(x).view(1, n_img * Lt, 1, C), dev(wqkv).view(3 * C, 1, 1, C), None, pad=0, ksize=1,
                     mma=L.MMA_BF16, wb=dev(wqkv).to(torch.bfloat16), out_bf16=True).view(n_img * Lt, 3 * C)
    assert qkv16.dtype == torch.bfloat16
    check(qkv16.float(), F.linear(bf16r(x), bf16r(wqkv)), 8e-3, 'linear bf16 out')
    o = torch.empty(n_img * Lt, C, device='cuda')
    s3 = (Lt * 3 * C, 3 * C, 0)
    ops.attention(qkv16, ops.offset(qkv16, C), ops.offset(qkv16, 2 * C), o, B=n_img * 4, H=1, Lq=Lt // 4, Lk=Lt // 4, D=C,
                  Dv=C, scale=1 / C ** 0.5, q_str=s3, k_str=s3, v_str=s3, o_str=(Lt * C, C, 0), mode=2, img_h=h8, img_w=w8,
                  ksplit=2, shift=4, kv_rot=P, n_img=n_img)
    q, k, v = qkv16.float().cpu().view(n_img, Lt, 3 * C).chunk(3, dim=-1)
    mask = O.shift_window_mask(h8, w8, h8 // 2, w8 // 2, h8 // 4, w8 // 4)
    kr, vr = torch.cat([k[P:], k[:P]]), torch.cat([v[P:], v[:P]])
    check(o.view(n_img, Lt, C), O._window_attention(q.contiguous(), kr, vr, 2, True, h8, w8, mask), 6e-3, 'bf16-in swin')
    Bc, T, Lt2, H, D = 2, 3, 64, 8, 48
    inner = H * D
    qkv = rnd('sq', (Bc * T, Lt2, 3 * inner))
    qd = dev(qkv).to(torch.bfloat16)
    o = torch.empty(Bc * T, Lt2, inner, device='cuda')
    s3 = (Lt2 * 3 * inner, 3 * inner, D)
    ops.attention(qd, ops.offset(qd, inner), ops.offset(qd, 2 * inner), o, B=Bc * T, H=H, Lq=Lt2, Lk=2 * Lt2, D=D, Dv=D,
                  scale=D ** -0.5, q_str=s3, k_str=s3, v_str=s3, o_str=(Lt2 * inner, inner, D), mode=1, T=T, seg_len=Lt2)
    qq, kk, vv = bf16r(qkv).chunk(3, dim=-1)
    former = torch.arange(T) - 1
    former[0] = 0

    def gather(t):
        t = t.reshape(Bc, T, Lt2, inner)
        return torch.cat([t[:, [0] * T], t[:, former]], dim=2).reshape(Bc * T, 2 * Lt2, inner)

    hs = lambda t: t.reshape(t.shape[0], t.shape[1], H, D).permute(0, 2, 1, 3)  # noqa: E731
    ref = ref_attn(hs(qq), hs(gather(kk)), hs(gather(vv)), D ** -0.5).permute(0, 2, 1, 3).reshape(Bc * T, Lt2, inner)
    check(o, ref, 6e-3, 'bf16-in sparse causal')


@pytest.mark.parametrize("cin,cout,hw,n,k,sk", [(512, 512, 16, 2, 3, None), (512, 512, 16, 1, 3, 3), (1024, 512, 16, 1, 1, 1),
                                                (384, 256, 8, 3, 1, None), (256, 64, 32, 1, 3, None)])
def test_conv_bf16_bk256_variant(cin, cout, hw, n, k, sk):
    """latency-bound small-M / deep-K layers: 256-channel K steps (single LDS buffer) must equal the default path."""
    x, w, b = rnd('kx', (n, cin, hw, hw)), rnd('kw', (cout, cin, k, k), 0.03), rnd('kb', (cout,))
    wp = pack(w)
    pro = ops.norm_affine(dev(nhwc(x)), None, None, cin, 1e-5)
    y = ops.conv(dev(nhwc(x)), wp, dev(b), pad=k // 2, ksize=k, mma=L.MMA_BF16, wb=wp.to(torch.bfloat16), split_k=sk,
                 pro=pro, pro_act=L.PRO_RELU)
    ref = F.conv2d(bf16r(F.relu(F.instance_norm(x, eps=1e-5))), bf16r(w), b, padding=k // 2)
    check(nchw(y), ref, 3e-3, f'bk256 {cin}->{cout} k{k} split {sk}')


def test_bilinear_upscale_matches_torch():
    """K0 (keep_arch.py:1020-1023): F.interpolate(scale_factor=4, mode='bilinear') on the device."""
    x = rnd('k0x', (2, 3, 24, 40))
    out = torch.empty((2, 3, 96, 160), device='cuda')
    L.call('keep_bilinear_upscale', dev(x), out, 6, 24, 40, 4)
    check(out, F.interpolate(x, scale_factor=4, mode='bilinear'), 1e-6, 'bilinear x4')


@pytest.mark.parametrize("mma", [L.MMA_F32, L.MMA_X3])
def test_plan_follows_a_retuned_tile_threshold(mma, monkeypatch):
    """The host sizes workspaces / statistics buffers from keep_conv2d_plan and mirrors no kernel internals: moving the
    library's small-M tile threshold through the environment changes the kernel instantiation, the split-K factor and
    the statistics layout, and the same Python call still produces the same numbers and consistent fused statistics."""
    x, w, b = rnd('rtx', (1, 128, 48, 48)), rnd('rtw', (192, 128, 1, 1), 0.1), rnd('rtb', (192,))
    wp = pack(w)
    kw = dict(pad=0, ksize=1, stats=True, mma=mma)
    if mma == L.MMA_X3:
        wx3, asc = x3w(wp)
        kw.update(wx3=wx3, x3_acc_scale=asc)
    ops.DEFAULT.profile = []
    y0, st0 = ops.conv(dev(nhwc(x)), wp, dev(b), **kw)
    # the plan sees 16 reference images x 48*48 = 36864 rows (batch-invariant plans): a "large" launch -> 128x128 tiles;
    # with the threshold above that it becomes a "small" one -> 64x64 tiles
    monkeypatch.setattr(ops.DEFAULT, 'flags', L.CONV_SMALL_TILES)
    y1, st1 = ops.conv(dev(nhwc(x)), wp, dev(b), **kw)
    names = [r[0] for r in ops.DEFAULT.profile]
    ops.DEFAULT.profile = None
    assert names[0] != names[1] and '2, 2, 2, 2' in names[0] and '2, 2, 1, 1' in names[1], names
    check(y1, y0, 2e-5, 'retuned tile')
    for y, st in ((y0, st0), (y1, st1)):
        if st is not None and st.part is not None:
            sc, sh = ops.norm_affine(y, None, None, 192, 1e-5, stats=st)
            sc2, sh2 = ops.norm_affine(y, None, None, 192, 1e-5)
            check(sc, sc2, 1e-5, 'stats scale'); check(sh, sh2, 1e-5, 'stats shift')
    if st0 is not None and st1 is not None and st0.part is not None and st1.part is not None:
        assert st0.P != st1.P          # 64-row vs 128-row partials: the layout moved and the host followed the plan


def test_conv_x3_halo_general_epilogue_two_blocks_per_cu():
    """The general (non-SIMPLE) epilogue of the x3 halo kernel -- activation, aux/CFT tensor, range-probed raw input -- on a
    launch that fills every CU with two blocks (N = 4 x 64^2 x 512 couts = 512 items).  Regression: a fully unrolled form
    of this epilogue stored a few wrong values per tile on exactly this launch shape while the small-N tests passed."""
    N, H, C = 4, 64, 256
    x = rnd('ge_x', (N, C, H, H), 30.0)
    w, b = rnd('ge_w', (2 * C, C, 3, 3), 0.05), rnd('ge_b', (2 * C,))
    wp = pack(w)
    wx3, asc = x3w(wp)
    for act in (L.ACT_LRELU02, L.ACT_RELU, L.ACT_NONE):
        y32 = ops.conv(dev(nhwc(x)), wp, dev(b), act=act)
        y3, st = ops.conv(dev(nhwc(x)), wp, dev(b), act=act, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, stats=True)
        check(y3, y32, 1e-5, f'x3 halo act={act}')
        assert torch.equal(st.amax.cpu(), y3.abs().flatten(1).max(1).values.cpu())
    # CFT form: residual + aux * cond on a channel slice of a wider input
    w2, b2 = rnd('ge_w2', (C, C, 3, 3), 0.05), rnd('ge_b2', (C,))
    w2p = pack(w2)
    w2x, asc2 = x3w(w2p)
    wide = dev(nhwc(rnd('ge_wide', (N, 2 * C, H, H), 20.0)))
    res, aux = dev(nhwc(rnd('ge_res', (N, C, H, H)))), dev(nhwc(rnd('ge_aux', (N, C, H, H))))
    for off in (0, C):
        kw = dict(cin=C, in_off=off, residual=res, aux=aux, aux_w=0.7)
        y32 = ops.conv(wide, w2p, dev(b2), **kw)
        y3 = ops.conv(wide, w2p, dev(b2), mma=L.MMA_X3, wx3=w2x, x3_acc_scale=asc2, **kw)
        check(y3, y32, 1e-5, f'x3 halo cft off={off}')


def test_conv_x3_halo_fused_probe_over_several_items_per_block():
    """The fused max|out| of the x3 halo kernel when a block walks several items of different images (768 items on 512 blocks): a wave
    goes to memory only above what it has already committed or seen for the image, and re-arms when the image changes -- the result
    equals a direct reduction of the output, per image, bit for bit."""
    N, H, C = 6, 128, 128
    x, w, b = rnd('fp_x', (N, C, H, H), 2.0), rnd('fp_w', (C, C, 3, 3), 0.05), rnd('fp_b', (C,))
    x = x * torch.tensor([1.0, 30.0, 0.01, 5.0, 0.3, 100.0]).view(N, 1, 1, 1)          # very different ranges per image
    gamma, beta = rnd('fp_g', (C,)) * 0.2 + 1, rnd('fp_bt', (C,)) * 0.2
    xd, wp, bd = dev(nhwc(x)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    res = dev(nhwc(rnd('fp_r', (N, C, H, H)))) * dev(torch.tensor([1.0, 50.0, 0.0, 2.0, 0.1, 1000.0])).view(N, 1, 1, 1)
    for pro in (True, False):
        kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, stats=True, residual=res)
        if pro:
            kw.update(pro=ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6), pro_act=L.PRO_SWISH)
        y, st = ops.conv(xd, wp, bd, **kw)
        assert torch.equal(st.amax.cpu(), y.abs().flatten(1).max(1).values.cpu()), f'fused max|out| (prologue={pro})'


def test_x3_scale_is_per_tensor():
    """engine/ops.py:make_x3_blob gives every tensor of a packed blob its own power-of-two scale (the ABI carries x3_acc_scale per
    launch): a huge weight in ONE tensor (here 1e6, seven decades above the others) must not cost another layer its `lo` halves.
    With one shared scale (rounds 2 / 3) the small tensor's `lo` falls into the fp16 subnormals and its convolution loses
    ~8 bits; per tensor, its error vs fp64 stays the exact-f32 kernel's."""
    cin, cout = 64, 64
    wa, wb_ = rnd('pt_a', (cout, 3, 3, cin), 0.05), rnd('pt_b', (cout, 3, 3, cin), 0.05)
    wb_[3, 1, 1, 5] = 1.0e6
    blob = dev(torch.cat([wa.reshape(-1), wb_.reshape(-1)]))
    index = {'a': (0, tuple(wa.shape)), 'b': (wa.numel(), tuple(wb_.shape))}
    views = {'a': blob[:wa.numel()].view(wa.shape), 'b': blob[wa.numel():].view(wb_.shape)}
    bx, table = ops.make_x3_blob(blob, index, views, ['a', 'b'])
    assert table[0][2] != table[1][2] and table[1][2] / table[0][2] >= 2.0 ** 20        # 2^-e: seven decades apart
    o = ops.Ops()
    o.set_precision(L.MMA_X3, blob, None, bx, 1.0, x3_scales=table)
    assert o.x3_scale_of(views['a']) == table[0][2] and o.x3_scale_of(views['b'][8:24]) == table[1][2]     # a row slice: its tensor's
    x = rnd('pt_x', (2, cin, 32, 32), 1.5)
    xd = dev(nhwc(x))
    y = o.conv(xd, views['a'], None, split_k=1)
    ref = F.conv2d(x.double(), wa.permute(0, 3, 1, 2).double(), None, padding=1).permute(0, 2, 3, 1)
    y32 = ops.conv(xd, views['a'], None, split_k=1)
    e3, e32 = err64(y, ref), err64(y32, ref)
    shared = ops.x3_scale_for(1.0e6)                                                      # what one scale for the whole blob would be
    y_sh = ops.conv(xd, views['a'], None, split_k=1, mma=L.MMA_X3, wx3=ops.split_x3(views['a'].reshape(-1, cin), shared).view(-1),
                    x3_acc_scale=1.0 / shared)
    e_sh = err64(y_sh, ref)
    print(f'per-tensor scale: x3 err {e3:.3e} (exact-f32 kernel {e32:.3e}); one shared scale: {e_sh:.3e}')
    assert e3 <= max(3.0 * e32, 2e-6) and e_sh > 10.0 * e3


@pytest.mark.parametrize("name,n,cin,cout,h,wd,up,res,gn", [('64->64 @512^2, GroupNorm-swish, residual', 16, 64, 64, 512, 512, False, True, True),
                                                            ('128->128 @256^2, GroupNorm-swish', 16, 128, 128, 256, 256, False, False, True),
                                                            ('128->64 @512^2, GroupNorm-swish', 16, 128, 64, 512, 512, False, False, True),
                                                            ('Upsample 128 @256^2 -> 512^2 (phase form)', 16, 128, 128, 256, 256, True, False, False)])
def test_conv_x3_at_the_shapes_of_the_step(name, n, cin, cout, h, wd, up, res, gn):
    """The launches that ARE the bench step (16 images: persistent blocks walk 4 ... 16 work items each, the streaming kernel's
    item seams and weight ring at full occupancy), against fp64 on sampled strips -- top and bottom border rows of the first and
    the last image and a strip in the middle of the batch -- with the exact-f32 kernel's error on the same strips as the yardstick."""
    g = torch.Generator().manual_seed(1234)
    x = (torch.randn((n, h, wd, cin), generator=g) * 2.0 + 0.3)
    w, b = rnd('rsw', (cout, cin, 3, 3), 0.05), rnd('rsb', (cout,))
    gamma, beta = rnd('rsg', (cin,)) * 0.2 + 1, rnd('rsbt', (cin,)) * 0.2
    Ho, Wo = (2 * h, 2 * wd) if up else (h, wd)
    r = torch.randn((n, Ho, Wo, cout), generator=g) if res else None
    xd, wp, bd = dev(x), pack(w), dev(b)
    wx3, asc = x3w(wp)
    kw = dict(upsample=up, stats=True, residual=None if r is None else dev(r))
    pro = None
    if gn:
        pro = ops.norm_affine(xd, dev(gamma), dev(beta), 32, 1e-6)
        kw.update(pro=pro, pro_act=L.PRO_SWISH)
    ops.DEFAULT.profile = []
    if up:         # the net's form of the Upsample convolutions: four 2x2-tap phase kernels (engine/ops.py:up2_phase_weights)
        w4 = ops.up2_phase_weights(wp)
        sc4 = ops.x3_scale_for(float(w4.abs().max()))
        y, st = ops.conv(xd, wp, bd, mma=L.MMA_X3, wx3=ops.split_x3(w4.reshape(-1, cin), sc4).view(-1), x3_acc_scale=1.0 / sc4,
                         **dict(kw, upsample=L.UPSAMPLE_X2_PHASES))
    else:
        y, st = ops.conv(xd, wp, bd, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, **kw)
    kname = ops.DEFAULT.profile[-1][0]
    ops.DEFAULT.profile = None
    assert kname == ('conv3x3_halo_x3_kernel<32, x2 phases>' if up else 'conv3x3_halo_x3s_kernel'), kname      # the streaming kernel IS what the step runs
    y32, _ = ops.conv(xd, wp, bd, **kw)
    assert torch.isfinite(y).all()
    worst3 = worst32 = 0.0
    scale = 1.0
    for (img, r0, r1) in ((0, 0, 12), (n // 2, Ho // 2 - 6, Ho // 2 + 6), (n - 1, Ho - 12, Ho)):
        # input rows the output rows r0 .. r1-1 need (on the upsampled grid: -1 .. +1), fp64 on the CPU
        lo, hi = max(r0 - 1, 0), min(r1 + 1, Ho)
        slo, shi = (lo // 2, (hi + 1) // 2) if up else (lo, hi)
        xs = x[img, slo:shi].double().permute(2, 0, 1)[None]                       # [1, C, rows, W]
        if gn:
            sc, sh = pro[0][img].double().cpu().view(1, cin, 1, 1), pro[1][img].double().cpu().view(1, cin, 1, 1)
            xs = xs * sc + sh
            xs = xs * torch.sigmoid(xs)
        if up:
            xs = F.interpolate(xs, scale_factor=2.0, mode='nearest')[:, :, lo - 2 * slo: lo - 2 * slo + (hi - lo)]
        top, bot = (1 if r0 == 0 else 0), (1 if r1 == Ho else 0)                  # zero padding only at the image border
        ref = F.conv2d(F.pad(xs, (1, 1, top, bot)), w.double(), b.double())
        assert ref.shape[2] == r1 - r0, (ref.shape, r0, r1)
        ref = ref[0].permute(1, 2, 0)
        if r is not None:
            ref = ref + r[img, r0:r1].double()
        worst3 = max(worst3, (y[img, r0:r1].double().cpu() - ref).abs().max().item())
        worst32 = max(worst32, (y32[img, r0:r1].double().cpu() - ref).abs().max().item())
        scale = max(scale, ref.abs().max().item())
    print(f'{name}: {kname}: x3 err {worst3:.3e}, exact-f32 kernel err {worst32:.3e} (scale {scale:.3g})')
    assert worst3 <= max(3.0 * worst32, 2e-6 * scale), f'x3 err {worst3:.3e} vs f32-kernel err {worst32:.3e} (scale {scale:.3g})'
    assert torch.equal(st.amax.cpu(), y.abs().flatten(1).max(1).values.cpu())
    sc1, sh1 = ops.norm_affine(y, None, None, 32, 1e-6, stats=st)
    sc2, sh2 = ops.norm_affine(y, None, None, 32, 1e-6)
    check(sc1, sc2, 1e-5, 'fused statistics: scale'); check(sh1, sh2, 1e-5, 'fused statistics: shift')


@pytest.mark.parametrize("n,cin,cout,h,wd,res", [(2, 64, 64, 32, 32, True), (3, 128, 128, 16, 64, False), (1, 32, 192, 8, 32, True),
                                                 (8, 32, 128, 64, 64, True)])      # the last: 1024 items on 512 blocks
def test_conv_x3_upsample_as_four_phase_convolutions(n, cin, cout, h, wd, res):
    """`upsample = KEEP_UPSAMPLE_X2_PHASES`: nearest x2 + 3x3 (VQ:146-156) as four 2x2-tap phase convolutions on the source grid
    (engine/ops.py:up2_phase_weights).  Against an fp64 reference of interpolate + conv2d the error is of the size of the 9-tap x3
    form's; the fused statistics and max|out| describe the scattered output; borders (phases looking outside the image) included."""
    x, w, b = rnd('u2x', (n, cin, h, wd), 2.0) + 0.3, rnd('u2w', (cout, cin, 3, 3), 0.05), rnd('u2b', (cout,))
    r = rnd('u2r', (n, cout, 2 * h, 2 * wd)) if res else None
    xd, wp, bd = dev(nhwc(x)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    w4 = ops.up2_phase_weights(wp)
    sc4 = ops.x3_scale_for(float(w4.abs().max()))
    w4x3 = ops.split_x3(w4.reshape(-1, cin), sc4).view(-1)
    kw = dict(stats=True, residual=None if r is None else dev(nhwc(r)), mma=L.MMA_X3)
    ops.DEFAULT.profile = []
    y2, st2 = ops.conv(xd, wp, bd, upsample=L.UPSAMPLE_X2_PHASES, wx3=w4x3, x3_acc_scale=1.0 / sc4, **kw)
    kname = ops.DEFAULT.profile[-1][0]
    ops.DEFAULT.profile = None
    assert 'phases' in kname, kname
    y1, _ = ops.conv(xd, wp, bd, upsample=True, wx3=wx3, x3_acc_scale=asc, **kw)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode='nearest'), w.double(), b.double(), padding=1)
    if r is not None:
        ref = ref + r.double()
    ref = ref.permute(0, 2, 3, 1)
    e2, e1 = err64(y2, ref), err64(y1, ref)
    scale = max(1.0, ref.abs().max().item())
    assert e2 <= max(3.0 * e1, 2e-6 * scale), f'phase form err {e2:.3e} vs 9-tap x3 err {e1:.3e} (scale {scale:.3g})'
    assert torch.equal(st2.amax.cpu(), y2.abs().flatten(1).max(1).values.cpu())
    sc, sh = ops.norm_affine(y2, None, None, cout, 1e-5, stats=st2)
    sc_d, sh_d = ops.norm_affine(y2, None, None, cout, 1e-5)
    check(sc, sc_d, 1e-5, 'phase form fused stats scale'); check(sh, sh_d, 1e-5, 'phase form fused stats shift')


def test_residual_in_place_matches_out_of_place():
    """`residual` may be the output buffer itself (y += conv(x)): the epilogues load their residual rows before the first store, and a
    thread reads exactly the addresses it later writes -- the in-place result equals the out-of-place one bit for bit on the x3 halo,
    x3 GEMM, exact-f32 halo and exact-f32 gather kernels."""
    N, H, C = 2, 64, 64
    x, w, b = rnd('ip_x', (N, C, H, H), 2.0), rnd('ip_w', (C, C, 3, 3), 0.05), rnd('ip_b', (C,))
    xd, wp, bd = dev(nhwc(x)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    r = dev(nhwc(rnd('ip_r', (N, C, H, H))))
    for kw in (dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc), dict()):
        ref = ops.conv(xd, wp, bd, residual=r, **kw)
        buf = r.clone()
        got = ops.conv(xd, wp, bd, residual=buf, out=buf, **kw)
        assert got.data_ptr() == buf.data_ptr() and torch.equal(got, ref), f'in-place residual, 3x3 ({"x3" if kw else "f32"})'
    tok = dev(rnd('ip_t', (4096, 256), 2.0))
    wl, bl = dev(rnd('ip_wl', (256, 256), 0.05)), dev(rnd('ip_bl', (256,)))
    wl3, ascl = x3w(wl.view(256, 1, 1, 256))
    rt = dev(rnd('ip_rt', (4096, 256)))
    for kw in (dict(mma=L.MMA_X3, wx3=wl3, x3_acc_scale=ascl), dict()):
        x4 = tok.view(1, 4096, 1, 256)
        ref = ops.conv(x4, wl.view(256, 1, 1, 256), bl, ksize=1, pad=0, residual=rt.view(1, 4096, 1, 256), **kw)
        buf = rt.clone().view(1, 4096, 1, 256)
        got = ops.conv(x4, wl.view(256, 1, 1, 256), bl, ksize=1, pad=0, residual=buf, out=buf, **kw)
        assert torch.equal(got, ref), f'in-place residual, GEMM ({"x3" if kw else "f32"})'


def test_conv_x3_rgb_first_conv_kernel():
    """KEEP_MMA_X3, Cin = 3 (VQ conv_in at 512^2): the im2col-in-LDS x3 kernel -- weights split on the fly, range-probed and
    bounded inputs, fused statistics and max|out| -- against the exact-f32 kernels."""
    x, w, b = rnd('c3x_x', (3, 3, 64, 96)), rnd('c3x_w', (64, 3, 3, 3), 0.2), rnd('c3x_b', (64,))
    wp = pack(w)
    for scale, bounded in ((1.0, True), (900.0, False)):
        xs = dev(nhwc(x * scale))
        ops.DEFAULT.profile = []
        y3, st = ops.conv(xs, wp, dev(b), mma=L.MMA_X3, wx3=None, x3_acc_scale=1.0, stats=True, bounded=bounded)
        name = ops.DEFAULT.profile[-1][0]
        ops.DEFAULT.profile = None
        assert name == 'conv3x3_c3_x3_kernel', name
        y32 = ops.conv(xs, wp, dev(b))
        ref64 = F.conv2d(x.double() * scale, w.double(), b.double(), padding=1)
        e3, e32 = err64(nchw(y3), ref64), err64(nchw(y32), ref64)
        assert e3 <= max(3.0 * e32, 2e-6 * float(ref64.abs().max())), (scale, e3, e32)
        assert torch.equal(st.amax.cpu(), y3.abs().flatten(1).max(1).values.cpu())
        sc, sh = ops.norm_affine(y3, None, None, 32, 1e-6, stats=st)
        sc2, sh2 = ops.norm_affine(y3, None, None, 32, 1e-6)
        check(sc, sc2, 1e-5, 'c3 x3 stats scale'); check(sh, sh2, 1e-5, 'c3 x3 stats shift')


def test_absmax_paths():
    """keep_absmax: contiguous rows (flat stream), a channel slice of wider rows, and the scalar path (C % 4 != 0); several images."""
    x = rnd('am_x', (3, 1000, 96), 5.0)
    x[1, 777, 13] = -123.5
    x[2, 3, 95] = 77.25
    xd = dev(x)
    assert torch.equal(ops.absmax(xd, 3, 1000, 96, 96, 1000 * 96).cpu(), x.abs().flatten(1).max(1).values)
    got = ops.absmax(xd.view(-1)[32:], 3, 1000, 48, 96, 1000 * 96).cpu()               # columns 32..79 of every row
    assert torch.equal(got, x[:, :, 32:80].abs().flatten(1).max(1).values)
    y = rnd('am_y', (2, 333, 7), 3.0)
    assert torch.equal(ops.absmax(dev(y), 2, 333, 7, 7, 333 * 7).cpu(), y.abs().flatten(1).max(1).values)
    big = rnd('am_big', (1, 300000, 128))
    assert torch.equal(ops.absmax(dev(big), 1, 300000, 128, 128, 300000 * 128).cpu(), big.abs().flatten(1).max(1).values)


@pytest.mark.parametrize("name,M,K,N,act,bias,two,img", [
    ('ffn1', 40000 + 77, 256, 1024, True, True, True, 0),        # GMFlow mlp.0 on cat(source, message): GELU, K-concatenated input, ragged M
    ('qkv', 300000 + 5, 128, 384, False, False, False, 0),       # GMFlow q/k/v projection of 4096-token maps
    ('ffn2', 270000, 1024, 128, False, True, False, 0),          # GMFlow mlp.2
    ('short', 4 * 65536, 128, 64, False, True, False, 4),        # VQGAN 1x1 shortcut 128 -> 64 at 256 x 256 with per-image range scales
    ('tail', 2048 * 128 + 1, 128, 100, False, True, False, 0)])  # Cout % 64 != 0: column tail, one row in the last row block
def test_linear_x3_at_the_gemm_shapes_of_the_step(name, M, K, N, act, bias, two, img):
    """conv_x3_kernel<.., ONE> (the row-major GEMM form of the x3 gather kernel) at the row counts and K / N of the step's token
    GEMMs and 1x1 shortcuts (tools/dev/conv_census.py), against float64: 2e-6 of sum |x| |w| per output."""
    x = rnd(f'gs_x_{name}', (M, K), 3.0 if img else 1.0)
    w = rnd(f'gs_w_{name}', (N, K), 0.05)
    b = rnd(f'gs_b_{name}', (N,), 0.3) if bias else None
    if img:
        x = x * torch.tensor([1.0, 40.0, 0.01, 7.0]).repeat_interleave(M // img).view(-1, 1)      # a different range per image
    wx3, asc = x3w(dev(w))
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pad=0, ksize=1, act=L.ACT_GELU if act else L.ACT_NONE, bounded=not img)
    xd, wd, bd = dev(x), dev(w), None if b is None else dev(b)
    if two:
        y = ops.conv(xd[:, :K // 2].contiguous().view(1, M, 1, K // 2), wd, bd, x2=xd[:, K // 2:].contiguous().view(1, M, 1, K // 2), **kw)
    else:
        y = ops.conv(xd.view(max(img, 1), M // max(img, 1), 1, K), wd, bd, **kw)
    ref = x.double() @ w.double().t() + (0 if b is None else b.double())
    if act:
        ref = torch.nn.functional.gelu(ref)
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert err64(y.reshape(M, N), ref) <= 2e-6 * scale, (name, err64(y.reshape(M, N), ref), scale)


@pytest.mark.parametrize("name,M,K,res,bias,img", [
    ('merge+norm1+res', 19 * 4096, 128, True, False, 0),      # GMFlow self-attention block: LN(merge(o)) + source
    ('merge+norm1', 6 * 4096, 128, False, True, 0),           # cross-attention block: LN(merge(o)) feeds the FFN
    ('mlp.2+norm2+res', 5 * 4096, 1024, True, True, 0),       # LN(mlp.2(h)) + source
    ('ranges', 4 * 4096, 128, True, True, 4)])                # per-image range scales on the GEMM input
def test_linear_x3_layernorm_epilogue(name, M, K, res, bias, img):
    """keep_conv2d ln_gamma (ABI v16): LayerNorm over the 128 output channels in the epilogue of the x3 GEMM form (tile <4,1,1,4>: one
    half-wave holds a whole row), applied BEFORE the residual -- GM/transformer.py:170-187.  Against float64 and against the two-launch
    form (GEMM, then keep_layernorm); refused loudly for any geometry it cannot run."""
    N = 128
    x = rnd(f'ln_x_{name}', (M, K), 3.0 if img else 1.0)
    w = rnd(f'ln_w_{name}', (N, K), 0.05)
    b = rnd(f'ln_b_{name}', (N,), 0.3) if bias else None
    g, be = 1.0 + 0.3 * rnd(f'ln_g_{name}', (N,)), rnd(f'ln_be_{name}', (N,), 0.2)
    r = rnd(f'ln_r_{name}', (M, N), 2.0) if res else None
    if img:
        x = x * torch.tensor([1.0, 40.0, 0.01, 7.0]).repeat_interleave(M // img).view(-1, 1)
    wd = dev(w)
    wx3, asc = x3w(wd)
    n_img = max(img, 1)
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pad=0, ksize=1, bounded=not img)
    xd, bd, gd, bed = dev(x).view(n_img, M // n_img, 1, K), None if b is None else dev(b), dev(g), dev(be)
    rd = None if r is None else dev(r).view(n_img, M // n_img, 1, N)
    y = ops.conv(xd, wd, bd, residual=rd, ln=(gd, bed, 1e-5), **kw).reshape(M, N)
    z = x.double() @ w.double().t() + (0 if b is None else b.double())
    ref = torch.nn.functional.layer_norm(z, (N,), g.double(), be.double(), 1e-5) + (0 if r is None else r.double())
    m = ops.conv(xd, wd, bd, **kw).reshape(M, N)
    two = ops.layernorm(m, gd, bed, res=None if r is None else dev(r))
    e_f, e_2 = err64(y, ref), err64(two, ref)
    print(f'{name}: fused {e_f:.3e}  two launches {e_2:.3e}  fused vs two {(y - two).abs().max().item():.3e}')
    assert e_f <= max(2.0 * e_2, 3e-6), (name, e_f, e_2)
    assert (y - two).abs().max().item() <= 1e-5
    with pytest.raises(L.KeepHipError):      # 256 output channels: a row does not fit one half-wave's tile
        w2 = dev(rnd('ln_w256', (256, K), 0.05))
        wx, a2 = x3w(w2)
        ops.conv(xd, w2, None, ln=(dev(rnd('ln_g256', (256,))), dev(rnd('ln_b256', (256,))), 1e-5),
                 **dict(kw, wx3=wx, x3_acc_scale=a2))
    with pytest.raises(L.KeepHipError):      # the exact-f32 policy has no LayerNorm epilogue
        ops.conv(xd, wd, bd, ln=(gd, bed, 1e-5), pad=0, ksize=1, bounded=True)


def test_rgb_s2d_and_the_4x4_form_of_gmflow_conv1():
    """keep_rgb_s2d: the normalised frame of keep_nchw_to_nhwc(mode 1), bit for bit, in the 2x2 space-to-depth layout (channel
    (dy*2 + dx)*3 + c, channels 12..15 zero); and GMFlow's 7x7 stride-2 convolution through it (4x4 stride-1, weights repacked by
    engine/weights.py:s2d_weights_7x7) against float64 under both parity policies."""
    from comfyui_keep_amd.engine.weights import s2d_weights_7x7
    N, H, W = 3, 64, 96
    x = rnd('s2d_x', (N, 3, H, W)).clamp(-1, 1)
    xd = dev(x)
    ref_img = ops.nchw_to_nhwc(xd, mode=1)                                           # [N,H,W,3]
    got = ops.rgb_s2d(xd)
    assert got.shape == (N, H // 2, W // 2, 16) and float(got[..., 12:].abs().max()) == 0.0
    want = ref_img.view(N, H // 2, 2, W // 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, 12)
    assert torch.equal(got[..., :12], want)
    w7 = rnd('s2d_w', (64, 3, 7, 7), 0.05)
    ws = dev(s2d_weights_7x7(w7.permute(0, 2, 3, 1).contiguous()))
    ref = F.conv2d(nchw(ref_img).double().cpu(), w7.double(), stride=2, padding=3).permute(0, 2, 3, 1)
    y32 = ops.conv(got, ws, None, stride=1, pad=2, ksize=4, out_hw=(H // 2, W // 2), split_k=1)
    wx3, asc = x3w(ws)
    y3 = ops.conv(got, ws, None, stride=1, pad=2, ksize=4, out_hw=(H // 2, W // 2), split_k=1, mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc,
                  bounded=True)
    e32, e3 = err64(y32, ref), err64(y3, ref)
    print(f's2d conv1: exact-f32 {e32:.3e}  x3 {e3:.3e}')
    assert y32.shape == (N, H // 2, W // 2, 64) and e32 <= 2e-5 and e3 <= max(3.0 * e32, 2e-6)


def test_linear_x3_k_concatenated_inputs():
    """keep_conv2d in2 (x3 GEMM form): cat([a, b], -1) @ W^T without materialising the concatenation (GM/transformer.py:182);
    equals the concat path bit for bit (same K order, same kernel), ragged M; rejected loudly outside the x3 policy."""
    M, C = 1000, 128
    a, b = dev(rnd('kc_a', (M, C))), dev(rnd('kc_b', (M, C), 2.0))
    w = dev(rnd('kc_w', (512, 2 * C), 0.05))
    wx3, asc = x3w(w)
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pad=0, ksize=1, act=L.ACT_GELU, bounded=True)
    y_cat = ops.conv(ops.concat2(a, b).view(1, M, 1, 2 * C), w, None, **kw)
    y_two = ops.conv(a.view(1, M, 1, C), w, None, x2=b.view(1, M, 1, C), **kw)
    assert torch.equal(y_cat, y_two)
    with pytest.raises(L.KeepHipError):
        ops.conv(a.view(1, M, 1, C), w, None, x2=b.view(1, M, 1, C), pad=0, ksize=1, bounded=True)      # exact-f32 policy


@pytest.mark.parametrize("M", [256, 777, 4096])
def test_gm_ffn_x3_fused_kernel(M):
    """keep_gm_ffn_x3 (round 5): LayerNorm(W2 . gelu(W0 . cat[src | msg])) + src in ONE launch, the [M, 1024] intermediate kept in
    registers (GM/transformer.py:137-142,182-187) -- against float64 (exact erf GELU) and against the two-launch form it replaces
    (x3 GEMM with the K-concatenated input + GELU epilogue, then x3 GEMM with the LayerNorm epilogue): its error vs float64 must not
    exceed twice the two-launch form's, and the two forms agree to 2e-5 (same products; the second GEMM sums its K axis in another
    order).  M = 777: a ragged last block (rows beyond M are neither read nor written)."""
    C, Hd = 128, 1024
    src, msg = rnd(f'ffn_src{M}', (M, C), 1.5), rnd(f'ffn_msg{M}', (M, C), 1.0)
    w0, w2 = rnd('ffn_w0', (Hd, 2 * C), 0.08), rnd('ffn_w2', (C, Hd), 0.05)
    g, be = 1.0 + 0.3 * rnd('ffn_g', (C,)), rnd('ffn_be', (C,), 0.2)
    x = torch.cat([src, msg], 1).double()
    h = x @ w0.double().t()
    h = 0.5 * h * (1.0 + torch.erf(h / 2.0 ** 0.5))
    ref = torch.nn.functional.layer_norm(h @ w2.double().t(), (C,), g.double(), be.double(), 1e-5) + src.double()
    sd, md, w0d, w2d, gd, bed = dev(src), dev(msg), dev(w0), dev(w2), dev(g), dev(be)
    wx0, a0 = x3w(w0d)
    w2p = ops.ffn_w2_perm(w2d)
    wx2p, a2p = x3w(w2p)
    guard = torch.full((64, C), 123.0, device='cuda')
    out = torch.cat([torch.empty((M, C), device='cuda'), guard])           # rows behind M must stay untouched
    L.call('keep_gm_ffn_x3', sd, md, wx0, float(a0), wx2p, float(a2p), gd, bed, 1e-5, out, M, C, Hd, 0)
    assert torch.equal(out[M:], guard)
    fused = out[:M]
    kw = dict(mma=L.MMA_X3, pad=0, ksize=1, bounded=True)
    if M % 128 == 0:
        wx2, a2 = x3w(w2d)
        hm = ops.conv(sd.view(1, M, 1, C), w0d, None, act=L.ACT_GELU, x2=md.view(1, M, 1, C), wx3=wx0, x3_acc_scale=a0, **kw)
        two = ops.conv(hm, w2d, None, residual=sd.view(1, M, 1, C), ln=(gd, bed, 1e-5), wx3=wx2, x3_acc_scale=a2, **kw).reshape(M, C)
        e_f, e_2 = err64(fused, ref), err64(two, ref)
        print(f'gm_ffn_x3 M={M}: fused {e_f:.3e}  two launches {e_2:.3e}  fused vs two {(fused - two).abs().max().item():.3e}')
        assert e_f <= max(2.0 * e_2, 5e-6), (e_f, e_2)
        assert (fused - two).abs().max().item() <= 2e-5
    else:
        e_f = err64(fused, ref)
        print(f'gm_ffn_x3 M={M}: fused {e_f:.3e}')
        assert e_f <= 2e-5
    # exact-erf variant (flags & KEEP_CONV_X3_EXACT_ACT)
    out2 = torch.empty((M, C), device='cuda')
    L.call('keep_gm_ffn_x3', sd, md, wx0, float(a0), wx2p, float(a2p), gd, bed, 1e-5, out2, M, C, Hd, 1)
    assert err64(out2, ref) <= 2e-5
    with pytest.raises(L.KeepHipError):
        L.call('keep_gm_ffn_x3', sd, md, wx0, float(a0), wx2p, float(a2p), gd, bed, 1e-5, out2, M, 64, Hd, 0)


def test_bgr_u8_to_comfy_equals_the_host_converter_on_every_value():
    """keep_bgr_u8_to_comfy == modules/utils.py:cv2_to_comfy_image (reference utils.py:162-166): RGB order, float32(u8) / 255 as one
    correctly rounded division -- all 256 values in all three channels, bit for bit (a tensor / scalar division in torch multiplies by
    the rounded reciprocal and misses the last bit on about half of them)."""
    from comfyui_keep_amd.modules.utils import cv2_to_comfy_image
    v = np.arange(256, dtype=np.uint8)
    img = np.stack([v, np.roll(v, 85), np.roll(v, 170)], -1).reshape(16, 16, 3)
    img = np.ascontiguousarray(np.concatenate([img, img[::-1, ::-1]], 0))
    out = torch.empty((32, 16, 3), dtype=torch.float32, device='cuda')
    L.call('keep_bgr_u8_to_comfy', torch.from_numpy(img).cuda(), out, 32 * 16)
    assert torch.equal(out.cpu(), cv2_to_comfy_image(img)[0])


def test_comfy_to_bgr_u8_equals_the_host_converter_on_every_boundary():
    """keep_comfy_to_bgr_u8 == modules/utils.py:comfy_image_to_cv2 (reference utils.py:155-160: `(x * 255).astype(np.uint8)` + RGB2BGR):
    every k / 255 and its two float32 neighbours (where the truncation flips), the values either side of every integer boundary k / 255
    as float32 candidates x with x * 255 within 2 ulp of k, and -- outside [0, 1] -- what numpy does on x86-64 (wrap through int32; NaN,
    inf and beyond-int32 -> 0).  Bit for bit, in all three channel positions, with a pixel count that is not a multiple of four."""
    from comfyui_keep_amd.modules.utils import comfy_image_to_cv2
    k = np.arange(257, dtype=np.float32)
    base = (k / np.float32(255.0)).astype(np.float32)
    cand = [base]
    for _ in range(3):
        cand.append(np.nextafter(cand[-1], np.float32(2.0)))
    lo = base
    for _ in range(3):
        lo = np.nextafter(lo, np.float32(-1.0))
        cand.append(lo)
    odd = np.array([0.0, -0.0, 1.0, 1.0039216, 1.01, 2.0, -0.001, -0.5, -1.0, 3.5, 8.5e6, 1e9, 3e9, -3e9, np.inf, -np.inf, np.nan,
                    1e-45, 0.9999999, 0.5, 0.49999997], np.float32)
    v = np.concatenate(cand + [odd, np.random.default_rng(0).random(4099, dtype=np.float32) * 1.2 - 0.1])
    v = v[: (v.size // 3) * 3]
    assert (v.size // 3) % 4 != 0
    for roll in range(3):
        img = np.roll(v, roll).reshape(1, -1, 3)                                    # [H=1, W, 3]
        with np.errstate(invalid='ignore'):
            ref = comfy_image_to_cv2(torch.from_numpy(img))
        out = torch.empty(img.shape, dtype=torch.uint8, device='cuda')
        L.call('keep_comfy_to_bgr_u8', torch.from_numpy(img).cuda(), out, img.shape[1])
        assert np.array_equal(out.cpu().numpy(), ref), roll


def test_frames_from_comfy_device_path_equals_the_host_converter():
    """modules/keep_processor.py:frames_from_comfy -- the node's IMAGE batch converted on the device by a worker thread in chunks (here 7
    frames in chunks of 2 + 2 + 2 + 1, and a non-contiguous batch) -- equals the per-frame host converter; frames and slices wait only for
    their chunks; KEEP_AMD_DEVICE_CONVERT=0 and a CPU device keep the host converter (plain list)."""
    from comfyui_keep_amd.modules import keep_processor as KP
    from comfyui_keep_amd.modules.utils import comfy_image_to_cv2
    g = torch.Generator().manual_seed(3)
    seq = torch.rand((7, 36, 50, 3), generator=g)
    ref = [comfy_image_to_cv2(seq[i]) for i in range(7)]
    fr = KP._ConvertedFrames(seq, torch.device('cuda', 0), chunk_bytes=2 * 36 * 50 * 3 * 4)
    assert fr._nchunks == 4 and len(fr) == 7
    assert np.array_equal(fr[6], ref[6]) and np.array_equal(fr[-7], ref[0])
    assert all(np.array_equal(a, b) for a, b in zip(fr[1:6], ref[1:6])) and all(np.array_equal(a, b) for a, b in zip(fr, ref))
    assert fr[0].dtype == np.uint8 and fr[0].flags['C_CONTIGUOUS']
    wide = torch.rand((5, 36, 50, 6), generator=g)[..., ::2]                         # a strided view
    assert not wide.is_contiguous()
    got = KP.frames_from_comfy(wide, 'cuda')
    assert isinstance(got, KP._ConvertedFrames) and all(np.array_equal(got[i], comfy_image_to_cv2(wide[i])) for i in range(5))
    assert isinstance(KP.frames_from_comfy(seq, 'cpu'), list)
    assert isinstance(KP.frames_from_comfy(seq.double(), 'cuda'), list)
    os.environ['KEEP_AMD_DEVICE_CONVERT'] = '0'
    try:
        assert isinstance(KP.frames_from_comfy(seq, 'cuda'), list)
    finally:
        del os.environ['KEEP_AMD_DEVICE_CONVERT']


@pytest.mark.parametrize("name,hw,K,N,act,bias,res,gn,ranged", [
    ('v / out_proj', 256, 512, 512, False, True, True, False, False),      # code transformer (KA:385-439): 512 -> 512 + residual
    ('linear1', 256, 512, 1024, True, True, False, False, False),          # 512 -> 1024 + GELU
    ('linear2', 256, 1024, 512, False, True, True, False, False),          # 1024 -> 512 + residual: two register groups per slice
    ('feat_emb', 256, 256, 512, False, True, False, False, False),         # 256 -> 512: four slices
    ('proj_out', 256, 512, 1536, False, True, True, False, False),         # wide N
    ('wide ff', 256, 2048, 512, False, True, False, False, True),          # K = 2048: four groups per slice, per-image range scale
    ('ragged rows', 192, 512, 2048, False, False, False, False, True)])
def test_gemm_x3_latency_form(name, hw, K, N, act, bias, res, gn, ranged):
    """gemm_x3l_kernel (keep_gemm_x3l.hip): the x3 GEMM form for <= 256 rows per image.  Against float64 (the bound of the
    throughput kernel: 2e-6 of sum |x| |w|), against conv_x3_kernel's one sequential sum (KEEP_CONV_NO_GEMM_LAT), and BIT-EQUAL
    between one image per launch (the latency kernel: one wave per K slice, LDS reduction) and 20 images per launch (conv_x3_kernel
    with canonical slices, ConvP.kslice_steps): the K slicing alone defines the sums, the launch follows the real row count."""
    n_img = 20
    M = n_img * hw
    x = rnd(f'gl_x_{name}', (n_img, hw, K), 2.0)
    if ranged:
        x = x * torch.tensor([1.0, 300.0, 0.01, 7.0, 2000.0]).repeat(4).view(-1, 1, 1)
    w = rnd(f'gl_w_{name}', (N, K), 0.05)
    b = rnd(f'gl_b_{name}', (N,), 0.3) if bias else None
    r = rnd(f'gl_r_{name}', (n_img, hw, N), 1.5) if res else None
    pro = None
    x_eff = x.double()
    if gn:
        sc, sh = rnd(f'gl_sc_{name}', (n_img, K)) * 0.3 + 1.0, rnd(f'gl_sh_{name}', (n_img, K)) * 0.2
        pro = (dev(sc), dev(sh))
        x_eff = x.double() * sc.double()[:, None] + sh.double()[:, None]
    wx3, asc = x3w(dev(w))
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pad=0, ksize=1, act=L.ACT_GELU if act else L.ACT_NONE, bounded=not ranged,
              pro=pro, stats=True)
    xd, wd, bd = dev(x), dev(w), None if b is None else dev(b)
    rd = None if r is None else dev(r).view(n_img, hw, 1, N)
    y, st = ops.conv(xd.view(n_img, hw, 1, K), wd, bd, residual=rd, **kw)
    pl = [k for k in ops._PLAN_CACHE.values() if k.kernel.startswith('gemm_x3l_kernel')]
    assert pl, 'the latency form was not selected'
    ref = x_eff.reshape(M, K) @ w.double().t() + (0 if b is None else b.double())
    if act:
        ref = torch.nn.functional.gelu(ref)
    if r is not None:
        ref = ref + r.double().reshape(M, N)
    scale = (x_eff.abs().reshape(n_img, hw, K) @ w.double().abs().t()).flatten(1).max(1).values       # per image: the ranges differ
    e = (y.double().cpu().reshape(n_img, hw * N) - ref.reshape(n_img, hw * N)).abs().max(1).values
    assert bool((e <= 2e-6 * scale).all()), (name, e / scale)
    # fused max|out| per image
    assert st is not None and st.amax is not None
    assert torch.equal(st.amax.cpu(), y.abs().reshape(n_img, -1).max(1).values.cpu())
    # one image per launch: same bits
    for i in (0, 7, 19):
        yi = ops.conv(xd[i:i + 1].view(1, hw, 1, K), wd, bd, residual=None if rd is None else rd[i:i + 1],
                      **{**kw, 'pro': None if pro is None else (pro[0][i:i + 1], pro[1][i:i + 1]), 'stats': False})
        assert torch.equal(yi.reshape(hw, N), y[i].reshape(hw, N)), (name, i)
    # the throughput kernel computes the same product in another order
    ops.DEFAULT.flags |= L.CONV_NO_GEMM_LAT
    y0 = ops.conv(xd.view(n_img, hw, 1, K), wd, bd, residual=rd, **{**kw, 'stats': False})
    e0 = (y0.double().cpu().reshape(n_img, hw * N) - ref.reshape(n_img, hw * N)).abs().max(1).values
    assert bool((e <= 2.0 * e0 + 1e-7 * scale).all()), (name, e / scale, e0 / scale)


@pytest.mark.parametrize("B,H,Lq,Lk,amp,peak", [(3, 8, 256, 256, 1.0, 3.0), (2, 4, 250, 200, 1.0, 3.0), (2, 8, 256, 256, 1e5, 3.0),
                                                (1, 2, 96, 64, 1.0, 60.0), (2, 3, 70, 37, 1.0, 3.0)])
def test_attention_x3_small_heads_latency_form(B, H, Lq, Lk, amp, peak):
    """attn_x3_small_kernel (D = Dv = 64, <= 256 keys: the code transformer's multi-head attention, KA:385-439): four waves x 64 keys
    per 32-query block, global row maximum through LDS, wave-ordered sum of the partial outputs.  Against float64 (not worse than
    attn_x3_kernel, the tile-by-tile online-softmax form it replaces), ragged Lq / Lk (masked keys, a wave without keys), packed q|k
    rows, probed ranges, peaked scores -- and bit-equal between one batch element per launch and all of them."""
    D = Dv = 64
    qk = rnd('asq', (B, max(Lq, Lk), 2 * H * D)) * amp                    # q | k packed like in_proj's output (strided heads)
    v = rnd('asv', (B, Lk, H * Dv)) * amp
    scale = D ** -0.5 * peak / amp
    q4, k4, v4 = qk[:, :Lq, :H * D].reshape(B, Lq, H, D), qk[:, :Lk, H * D:].reshape(B, Lk, H, D), v.reshape(B, Lk, H, Dv)
    ref = torch.softmax(torch.einsum('bqhd,bkhd->bhqk', q4.double(), k4.double()) * scale, -1)
    ref = torch.einsum('bhqk,bkhd->bqhd', ref, v4.double())
    Lm = max(Lq, Lk)
    qkd, vd = dev(qk), dev(v)

    def run(flags, b0=0, nb=B):
        o = torch.empty((nb, Lq, H, Dv), device='cuda')
        ops.DEFAULT.attn_flags = flags
        x = qkd[b0:b0 + nb]
        ops.attention(x, ops.offset(x, H * D), vd[b0:b0 + nb], o, B=nb, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=scale,
                      q_str=(Lm * 2 * H * D, 2 * H * D, D), k_str=(Lm * 2 * H * D, 2 * H * D, D), v_str=(Lk * H * Dv, H * Dv, Dv),
                      o_str=(Lq * H * Dv, H * Dv, Dv), mma=L.MMA_X3, probe=amp > 1)
        return o
    o_new, o_old = run(0), run(L.ATTN_NO_SMALL)
    assert torch.isfinite(o_new).all()
    sc = ref.abs().max().item()
    e_new, e_old = err64(o_new, ref), err64(o_old, ref)
    assert e_new <= max(2.0 * e_old, 2e-6 * sc), f'latency form {e_new:.3e} vs attn_x3_kernel {e_old:.3e} (scale {sc:.3g})'
    assert not torch.equal(o_new, o_old) or Lq < 64, 'both flag settings ran the same kernel'
    for b0 in range(B):
        assert torch.equal(run(0, b0, 1)[0], o_new[b0]), f'batch element {b0} differs between B = 1 and B = {B}'


def test_attention_x3_two_pass_latency_form():
    """attn_scores_x3l_kernel + attn_pv_x3l_kernel (the VQGAN AttnBlock on a 16 x 16 map, VQ:219-243: d = 512, 256 tokens) on q|k|v packed
    in one [tokens, 3C] buffer like the qkv projection writes them: against float64 (not worse than attn_x3_sfull2_kernel), peaked
    scores, and bit-equal between one image per launch and all of them."""
    B, H, L_, D = 3, 1, 256, 512
    for peak in (3.0, 40.0):
        qkv = rnd(f'tp_qkv{peak}', (B, L_, 3 * D))
        q4, k4, v4 = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        scale = D ** -0.5 * peak
        ref = torch.softmax(torch.einsum('bqd,bkd->bqk', q4.double(), k4.double()) * scale, -1) @ v4.double()
        qd = dev(qkv)

        def run(flags, b0=0, nb=B):
            o = torch.empty((nb, L_, D), device='cuda')
            ops.DEFAULT.attn_flags = flags
            x = qd[b0:b0 + nb]
            s3 = (L_ * 3 * D, 3 * D, 0)
            ops.attention(x, ops.offset(x, D), ops.offset(x, 2 * D), o, B=nb, H=H, Lq=L_, Lk=L_, D=D, Dv=D, scale=scale,
                          q_str=s3, k_str=s3, v_str=s3, o_str=(L_ * D, D, 0), mma=L.MMA_X3)
            return o
        o_new, o_old = run(0), run(L.ATTN_NO_TWO_PASS)
        assert torch.isfinite(o_new).all()
        sc = ref.abs().max().item()
        e_new, e_old = err64(o_new, ref), err64(o_old, ref)
        assert e_new <= max(2.0 * e_old, 2e-6 * sc), f'two passes {e_new:.3e} vs attn_x3_sfull2_kernel {e_old:.3e} (scale {sc:.3g})'
        assert not torch.equal(o_new, o_old), 'both flag settings ran the same kernel'
        for b0 in range(B):
            assert torch.equal(run(0, b0, 1)[0], o_new[b0]), f'batch element {b0} differs between B = 1 and B = {B}'


@pytest.mark.parametrize("n,h,wd,cin,cout,sk,gn,res,ranged", [
    (1, 16, 16, 512, 512, None, True, True, False),      # the 16 x 16 ResBlock convolutions of the frame recurrence (plan: split 4)
    (2, 16, 16, 256, 512, 4, False, False, True),        # no prologue: probed range scale per image
    (1, 16, 16, 1024, 512, None, True, False, False),    # CFT encode_enc conv1 on cat[enc, dec]
    (1, 32, 32, 256, 256, 4, True, True, False),         # a wide map under a latency-profile split
    (1, 16, 16, 80, 64, 2, True, False, False),          # five chunks over two splits: uneven channel ranges
    (1, 20, 16, 64, 128, 2, False, True, False)])        # 20 rows: five 4-row tiles, a map the 256-pixel kernel cannot tile -> never selected
def test_conv_x3_small_tile_partials_are_the_256_pixel_kernels(n, h, wd, cin, cout, sk, gn, res, ranged):
    """conv3x3_x3p_kernel (keep_conv_x3p.hip): the split-K partials of a 3x3 x3 convolution from 64-pixel blocks when few images are in
    flight -- BIT-EQUAL to conv3x3_halo_x3_kernel's 256-pixel blocks (KEEP_CONV_NO_SMALL_PARTIALS): same channel ranges, same prologue
    arithmetic, same MFMA order, same reducer."""
    if (h * wd) % 256:
        pytest.skip('not a map of the halo path')
    x = rnd('sp_x', (n, cin, h, wd), 2.0) + 0.3
    if ranged:
        x = x * torch.tensor([1.0, 900.0][:n]).view(-1, 1, 1, 1)
    w, b = rnd('sp_w', (cout, cin, 3, 3), 0.05), rnd('sp_b', (cout,))
    r = rnd('sp_r', (n, cout, h, wd)) if res else None
    xd, wp, bd = dev(nhwc(x)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    pro = None
    if gn:
        pro = ops.norm_affine(xd, dev(rnd('sp_g', (cin,)) * 0.2 + 1), dev(rnd('sp_bt', (cin,)) * 0.2), 16, 1e-6)
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pro=pro, pro_act=L.PRO_SWISH if gn else L.PRO_NONE, split_k=sk,
              residual=None if r is None else dev(nhwc(r)), bounded=not ranged)
    outs = []
    for fl in (0, L.CONV_NO_SMALL_PARTIALS):
        ops.DEFAULT.flags = fl
        outs.append(ops.conv(xd, wp, bd, **kw))
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
    # and it is a convolution (fp32-grade)
    xe = x.double()
    if gn:
        xe = xe * pro[0].cpu().double().view(n, cin, 1, 1) + pro[1].cpu().double().view(n, cin, 1, 1)
        xe = xe * torch.sigmoid(xe)
    ref = F.conv2d(xe, w.double(), b.double(), padding=1) + (0 if r is None else r.double())
    scale = F.conv2d(xe.abs(), w.double().abs(), padding=1).flatten(1).max(1).values.view(n, 1, 1, 1)
    assert bool(((nchw(outs[0]).double().cpu() - ref).abs() <= 3e-6 * scale).all())


@pytest.mark.parametrize("n,h,wd,cin,cout,gn,res,ranged,act,up", [
    (1, 32, 32, 256, 256, True, True, False, False, False),       # ResBlock convolutions of the 32 x 32 stage (16 chunks)
    (1, 64, 64, 256, 256, True, True, False, False, False),       # 64 x 64 stage: 64 items of the streaming kernel -> 1024 small blocks
    (1, 32, 32, 512, 256, True, False, False, False, False),      # 32 chunks
    (1, 64, 64, 128, 256, True, True, False, False, False),       # 8 chunks
    (2, 32, 32, 256, 512, False, False, True, False, False),      # no prologue, probed range scale per image
    (1, 64, 64, 144, 256, False, True, False, False, False),      # 9 chunks: the loop form of the ring
    (1, 32, 64, 64, 64, True, False, False, False, False),        # 4 chunks, a non-square map
    (1, 32, 32, 256, 512, False, False, True, True, False),       # round 6: CFT scale.0|shift.0 + LeakyReLU(0.2) (KA:468-469), probed input
    (1, 64, 64, 256, 512, False, False, True, True, False),
    (1, 16, 16, 512, 512, False, False, True, False, True),       # round 6: the 16 -> 32 Upsample in its 9-tap form (nearest x2 in the halo addresses)
    (2, 16, 32, 128, 64, False, True, True, True, True)])         # both, two images, a non-square source
def test_conv_x3_small_tiles_equal_the_streaming_kernel(n, h, wd, cin, cout, gn, res, ranged, act, up):
    """conv3x3_x3q_kernel + conv_stats_replica_kernel (un-split plans on wide maps, few items): output, GroupNorm partials and max|out|
    BIT-EQUAL to conv3x3_halo_x3s_kernel's (KEEP_CONV_NO_SMALL_PARTIALS) -- its conversion arithmetic, product order and epilogue order
    (round 6: with an epilogue activation and with the nearest-x2 source addressing too)."""
    x = rnd('sq_x', (n, cin, h, wd), 2.0) + 0.3
    if ranged:
        x = x * torch.tensor([1.0, 900.0][:n]).view(-1, 1, 1, 1)
    w, b = rnd('sq_w', (cout, cin, 3, 3), 0.05), rnd('sq_b', (cout,))
    r = rnd('sq_r', (n, cout, (2 * h if up else h), (2 * wd if up else wd))) if res else None
    xd, wp, bd = dev(nhwc(x)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    pro = None
    if gn:
        pro = ops.norm_affine(xd, dev(rnd('sq_g', (cin,)) * 0.2 + 1), dev(rnd('sq_bt', (cin,)) * 0.2), 16, 1e-6)
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, pro=pro, pro_act=L.PRO_SWISH if gn else L.PRO_NONE,
              residual=None if r is None else dev(nhwc(r)), bounded=not ranged, stats=True, split_k=1,
              act=L.ACT_LRELU02 if act else L.ACT_NONE, upsample=bool(up))
    old_up2 = ops.UP2_PHASES
    ops.UP2_PHASES = False          # (the 9-tap form is what the net runs at 16 -> 32: 16-wide source tiles)
    got = []
    for fl in (0, L.CONV_NO_SMALL_PARTIALS):
        ops.DEFAULT.flags = fl
        ops.DEFAULT.amax_arena = None          # fresh (zeroed by the library) max|out| slots per call
        y, st = ops.conv(xd, wp, bd, **kw)
        got.append((y.clone(), st.part.clone(), st.amax.clone()))
    ops.UP2_PHASES = old_up2
    assert torch.isfinite(got[0][0]).all()
    assert torch.equal(got[0][0], got[1][0]), 'outputs differ'
    assert torch.equal(got[0][1], got[1][1]), 'GroupNorm partials differ'
    assert torch.equal(got[0][2], got[1][2]), 'max|out| differs'
    assert torch.equal(got[0][2].cpu(), got[0][0].abs().flatten(1).max(1).values.cpu())
    if act or up:      # ... and the value itself against float64 (the bound of the kernel family: 3e-6 of sum |x| |w|)
        xe = nchw(xd).double().cpu()
        if up:
            xe = F.interpolate(xe, scale_factor=2.0, mode='nearest')
        ref = F.conv2d(xe, w.double(), b.double(), padding=1)
        if act:
            ref = F.leaky_relu(ref, 0.2)
        if r is not None:
            ref = ref + r.double()
        scale = F.conv2d(xe.abs(), w.double().abs(), padding=1).flatten(1).max(1).values.view(n, 1, 1, 1)
        assert bool(((nchw(got[0][0]).double().cpu() - ref).abs() <= 3e-6 * scale).all())


@pytest.mark.parametrize("n,h,wd,cin,cout,act,aux,res,stats,slice_in", [
    (1, 32, 32, 256, 256, False, True, True, True, True),        # CFT shift convolution (KA:470-472): channel slice in, residual + aux, statistics
    (1, 64, 64, 256, 256, False, True, True, True, True),
    (1, 16, 16, 512, 1024, True, False, False, 'amax', False),   # CFT scale.0|shift.0 + LeakyReLU at 16 x 16: 16-wide tiles, only max|out|
    (2, 16, 16, 512, 512, False, False, True, 'amax', False),    # a plain 16-wide convolution with a residual, two images
    (1, 32, 32, 128, 64, True, True, True, True, False)])        # activation AND aux
def test_conv_x3_small_tiles_full_epilogue_equal_the_stage_barrier_kernel(n, h, wd, cin, cout, act, aux, res, stats, slice_in):
    """Round 6: un-split plans the streaming kernel does not take (aux tensor, 16-wide maps) with few items -- conv3x3_x3p_kernel's blocks with
    the full epilogue (+ conv_stats_replica_kernel) -- BIT-EQUAL to conv3x3_halo_x3_kernel (KEEP_CONV_NO_SMALL_PARTIALS): output, GroupNorm
    partials, max|out|; and the value against float64."""
    ld = 2 * cin if slice_in else cin
    xfull = rnd('sf_x', (n, ld, h, wd), 2.0) + 0.3
    w, b = rnd('sf_w', (cout, cin, 3, 3), 0.05), rnd('sf_b', (cout,))
    r = rnd('sf_r', (n, cout, h, wd)) if res else None
    a_t = rnd('sf_a', (n, cout, h, wd)) if aux else None
    xd, wp, bd = dev(nhwc(xfull)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, residual=None if r is None else dev(nhwc(r)), aux=None if a_t is None else dev(nhwc(a_t)),
              aux_w=0.75, act=L.ACT_LRELU02 if act else L.ACT_NONE, stats=stats, split_k=1)
    if slice_in:
        kw.update(cin=cin, in_off=cin)
    got = []
    for fl in (0, L.CONV_NO_SMALL_PARTIALS):
        ops.DEFAULT.flags = fl
        ops.DEFAULT.amax_arena = None
        y, st = ops.conv(xd, wp, bd, **kw)
        got.append((y.clone(), None if st.part is None else st.part.clone(), st.amax.clone()))
    ops.DEFAULT.flags = 0
    assert torch.isfinite(got[0][0]).all()
    assert torch.equal(got[0][0], got[1][0]), 'outputs differ'
    assert (got[0][1] is None) == (got[1][1] is None) and (got[0][1] is None or torch.equal(got[0][1], got[1][1])), 'GroupNorm partials differ'
    assert torch.equal(got[0][2], got[1][2]), 'max|out| differs'
    xe = xfull[:, cin:] if slice_in else xfull
    ref = F.conv2d(xe.double(), w.double(), b.double(), padding=1)
    if act:
        ref = F.leaky_relu(ref, 0.2)
    if r is not None:
        ref = (r.double() + 0.75 * (r.double() * a_t.double() + ref)) if aux else ref + r.double()
    scale = F.conv2d(xe.double().abs(), w.double().abs(), padding=1).flatten(1).max(1).values.view(n, 1, 1, 1)
    assert bool(((nchw(got[0][0]).double().cpu() - ref).abs() <= 4e-6 * scale * (1.75 if aux else 1.0)).all())


@pytest.mark.parametrize("n,h,wd,cin,cout,down", [(1, 64, 64, 256, 256, True), (1, 32, 32, 256, 256, True), (1, 128, 128, 128, 128, True),
                                                  (2, 64, 64, 128, 256, True), (3, 32, 32, 128, 128, True), (1, 32, 32, 256, 128, False)])
def test_conv_x3_gather_small_tile_with_statistics_replica(n, h, wd, cin, cout, down):
    """The encoder's stride-2 convolutions with few rows in flight: conv_x3_kernel's 64 x 64 tile + conv_x3_gather_stats_replica_kernel
    give the output, the GroupNorm partials (the plan's 128-row partition) and max|out| of the 128 x 128 tile's fused epilogue
    (KEEP_CONV_NO_SMALL_PARTIALS), bit for bit; a 1x1 GEMM-form launch with statistics takes the same route."""
    x = rnd('gs_x', (n, cin, h, wd), 2.0) + 0.2
    k = 3 if down else 1
    w, b = rnd('gs_w', (cout, cin, k, k), 0.05), rnd('gs_b', (cout,))
    xd, wp, bd = dev(nhwc(x)), pack(w), dev(b)
    wx3, asc = x3w(wp)
    kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=asc, bounded=True, stats=True, split_k=1)
    kw.update(dict(down=True) if down else dict(ksize=1, pad=0))
    got = []
    for fl in (0, L.CONV_NO_SMALL_PARTIALS):
        ops.DEFAULT.flags = fl
        ops.DEFAULT.amax_arena = None
        y, st = ops.conv(xd, wp, bd, **kw)
        got.append((y.clone(), st.part.clone(), None if st.amax is None else st.amax.clone(), st.P))
    ops.DEFAULT.flags = 0
    hw_o = got[0][0].shape[1] * got[0][0].shape[2]
    P = got[0][3]
    assert P == got[1][3] and P in (hw_o // 128, hw_o // 64), (P, hw_o)      # the plan's partition (reference batch of 16): 128 rows, 64 on the smallest maps
    assert torch.isfinite(got[0][0]).all()
    assert torch.equal(got[0][0], got[1][0]), 'outputs differ'
    assert torch.equal(got[0][1], got[1][1]), 'GroupNorm partials differ'
    assert (got[0][2] is None) == (got[1][2] is None) and (got[0][2] is None or torch.equal(got[0][2], got[1][2])), 'max|out| differs'
    ref = F.conv2d(F.pad(x.double(), (0, 1, 0, 1)) if down else x.double(), w.double(), b.double(), stride=2 if down else 1)
    assert (nchw(got[0][0]).cpu().double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    part = got[0][1].double()                                                # [N, P, C, 2]: sums / sums of squares per partial
    yy = got[0][0].double().flatten(1, 2).view(n, P, hw_o // P, cout)
    assert (part[..., 0] - yy.sum(2)).abs().max() <= 1e-3 and (part[..., 1] - (yy * yy).sum(2)).abs().max() <= 1e-3 * max(1.0, float((yy * yy).sum(2).max()))


def test_layernorm_and_geglu_with_fused_range_maxima():
    """keep_layernorm_amax / keep_geglu_amax: outputs bit-identical to keep_layernorm / keep_geglu, amax[n] = max |out| over image n's
    rows (what keep_absmax returns for the output), with caller-zeroed slots and without; rows per image 1 .. 1024, ragged C."""
    from comfyui_keep_amd.engine import ops
    o = ops.Ops()
    o.begin_forward(torch.device('cuda'))
    for (n_img, rows, C, with_res) in ((1, 256, 512, True), (3, 64, 256, True), (2, 1024, 256, False), (5, 7, 100, True), (48, 16, 128, False)):
        x = op_input(f'lnamax_x_{C}_{rows}', (n_img * rows, C), 3.0).cuda() * torch.linspace(0.5, 40.0, n_img).repeat_interleave(rows).view(-1, 1).cuda()
        g, b = op_input(f'lnamax_g_{C}', (C,)).cuda() + 1.0, op_input(f'lnamax_b_{C}', (C,)).cuda()
        res = op_input(f'lnamax_r_{C}_{rows}', (n_img * rows, C), 5.0).cuda() if with_res else None
        ref = ops.Ops.layernorm(x, g, b, res=res)
        y, am = o.layernorm_amax(x, g, b, res=res, n_img=n_img)
        assert torch.equal(y, ref) and torch.equal(am, ref.view(n_img, -1).abs().max(1).values), (n_img, rows, C)
        am2 = torch.full((n_img,), 7e30, device='cuda')           # not zeroed: the entry point clears the slots itself
        y2 = torch.empty_like(x)
        L.call('keep_layernorm_amax', x, g, b, res, y2, n_img * rows, C, 1e-5, rows, am2, 0)
        assert torch.equal(y2, ref) and torch.equal(am2, am)
    for (n_img, rows, F) in ((1, 256, 1024), (3, 64, 2048), (4, 5, 36), (48, 256, 1024)):
        x = op_input(f'ggamax_{rows}_{F}', (n_img * rows, 2 * F), 2.0).cuda() * torch.linspace(0.3, 20.0, n_img).repeat_interleave(rows).view(-1, 1).cuda()
        ref = ops.Ops.geglu(x)
        y, am = o.geglu_amax(x, n_img=n_img)
        assert torch.equal(y, ref) and torch.equal(am, ref.view(n_img, -1).abs().max(1).values), (n_img, rows, F)
        am2 = torch.full((n_img,), 7e30, device='cuda')
        y2 = torch.empty_like(ref)
        L.call('keep_geglu_amax', x, y2, n_img, rows, F, am2, 0)
        assert torch.equal(y2, ref) and torch.equal(am2, am)
    with pytest.raises(L.KeepHipError):
        L.call('keep_layernorm_amax', x, g, b, None, y2, 10, 128, 1e-5, 3, am2, 0)


@pytest.mark.parametrize('hw,cin,cout,swish', [(32, 64, 64, True), (48, 64, 64, False), (64, 128, 128, True)])
def test_winograd_x3_microkernel_vs_fp64(hw, cin, cout, swish):
    """tools/dev/winograd_probe.hip (evidence for DESIGN 5.7, NOT product): the Winograd F(2x2, 3x3) x3 microkernel -- GroupNorm affine (+ swish) prologue, bias
    epilogue, persistent blocks, ragged tile counts per block -- against an fp64 convolution of the same activated input: fp32-grade like the product's x3 kernels."""
    import sys
    dev = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'dev')
    if not os.path.exists(os.path.join(dev, 'libwinograd_probe.so')):
        pytest.skip('tools/dev/libwinograd_probe.so not built (python __graft_entry__.py)')
    sys.path.insert(0, dev)
    import winograd_probe as WP
    lib = WP.load('')
    torch.manual_seed(hw + cin)
    N = 3
    w = (torch.randn(cout, 3, 3, cin) * 0.05).cuda()
    bias = (torch.randn(cout) * 0.1).cuda()
    scale, shift = (torch.rand(N, cin) + 0.5).cuda(), (torch.randn(N, cin) * 0.2).cuda()
    x = torch.randn(N, hw, hw, cin, device='cuda')
    u, inv = WP.pack_weights(w)
    y = WP.run(lib, x, u, bias, scale, shift, inv, swish=swish)
    a = x.double() * scale.double()[:, None, None, :] + shift.double()[:, None, None, :]
    if swish:
        a = a * torch.sigmoid(a)
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1)
    err = (y.double() - ref).abs().max().item()
    assert torch.isfinite(y).all() and err <= 4e-6 * max(1.0, ref.abs().max().item()), err
