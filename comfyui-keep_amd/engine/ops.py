"""Tensor-level wrappers over the C-ABI: shape bookkeeping + output allocation only.

All activations are channels-last fp32 device tensors: feature maps ``[N,H,W,C]``, token
matrices ``[M,C]`` (same memory).  torch is used for allocation (caching allocator -> safe
inside hipGraph capture) and views; every arithmetic op is a ``libkeep_hip.so`` kernel.
"""
import math
import os

import torch

from . import hiplib as L

# below this many matrix-core waves a conv launch cannot fill 256 CUs x 4 SIMDs -> split K
_TARGET_WAVES = 1024

# matrix-core operand precision of keep_conv2d launches (L.MMA_F32 = parity policy, L.MMA_BF16 = speed policy) and the
# weight blobs the bf16 twins of fp32 weight views are resolved from (same element offsets in both blobs)
MMA = L.MMA_F32
ATTN_MMA = L.MMA_F32
_BLOB32 = None
_BLOB16 = None


def set_precision(mma, blob32=None, blob16=None):
    global MMA, ATTN_MMA, _BLOB32, _BLOB16
    MMA, ATTN_MMA, _BLOB32, _BLOB16 = mma, mma, blob32, blob16


def bf16_twin(w):
    """bf16 copy of an fp32 weight view that lives inside the registered packed blob."""
    if _BLOB32 is None or _BLOB16 is None:
        raise RuntimeError("bf16 policy needs the packed weight blobs registered (ops.set_precision)")
    off = (w.data_ptr() - _BLOB32.data_ptr()) // 4
    if off < 0 or off + w.numel() > _BLOB32.numel() or not w.is_contiguous():
        raise RuntimeError("weight view is not inside the packed blob; pass wb= explicitly")
    return _BLOB16[off:off + w.numel()]


C3 = os.environ.get('KEEP_NO_C3') is None                  # dev switch: Cin <= 3 first convs on the gather kernel
COUT4 = os.environ.get('KEEP_NO_COUT4') is None            # dev switch: Cout <= 4 3x3 convs on the gather kernel
TOKEN_LINEAR = os.environ.get('KEEP_NO_TOKEN_LINEAR') is None   # dev switch: streaming GEMM for the GMFlow projections
HALO_F32 = os.environ.get('KEEP_NO_HALO_F32') is None     # dev switch: fall back to the gather kernel
HALO_V1 = os.environ.get('KEEP_HALO_VER', '3') == '1'   # halo kernel generation (3 = persistent, default)
USE_BK256 = bool(int(os.environ.get('KEEP_BK256', '0')))   # measured slower than BK=64 + split-K at B<=4 (kept for A/B)
# two-pass normalise+activate -> bf16 in front of the halo conv for inputs of at least this many pixels per launch
# (N*H*W); below it the halo kernel applies the affine + activation itself while staging (measured crossover)
HALO_PRENORM_MINPIX = int(os.environ.get('KEEP_HALO_PRENORM_MINPIX', '0'))

# bench.py's roofline leg: when a list, every keep_conv2d launch is bracketed by HIP events on the launch stream
# and appended as (kernel instantiation, algorithmic_flops, split_k, start_event, end_event, algorithmic_bytes)
PROFILE = None


def tile_config(M, Cout, bf16=False, bk256=False, plain=False):
    """Name of the gather-kernel instantiation keep_conv.hip selects, as rocprofv3 prints it (keep in sync)."""
    t = '4, 1, 1, 1' if Cout <= 32 else ('2, 2, 1, 1' if (Cout <= 64 or M <= 4096) else '2, 2, 2, 2')
    if bf16:
        plain = plain and not bk256 and t != '4, 1, 1, 1'
        return f"conv_bf16_kernel<{t}, {'256, 1' if bk256 else '64, 1'}, {'true' if plain else 'false'}>"
    return f'conv_f32_kernel<{t}>'


def empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _tile_blocks(M, Cout):
    """Mirror of the tile selection in keep_conv.hip (keep in sync)."""
    if Cout <= 32:
        return math.ceil(M / 128) * math.ceil(Cout / 32)
    if Cout <= 64 or M <= 4096:
        return math.ceil(M / 64) * math.ceil(Cout / 64)
    return math.ceil(M / 128) * math.ceil(Cout / 128)


def pick_split_k(M, Cout, nsteps, bf16=False):
    waves = _tile_blocks(M, Cout) * 4
    if waves >= _TARGET_WAVES or nsteps < 8:
        return 1
    # small layers are latency-bound (one global round trip per K step): oversubscribe the CUs 4x so that several
    # blocks per CU overlap their loads, keeping >= 2 K steps per split
    if bf16:
        s = min(4 * _TARGET_WAVES // waves, nsteps // 2, 32)
    else:
        s = min(_TARGET_WAVES // waves, nsteps // 4, 32)
    return max(1, s)


def halo_bf16_eligible(Cin, Cout, Ho, Wo, ld=None, in_off=0):
    """Geometry accepted by the persistent bf16 LDS-halo kernel (3x3, stride 1, pad 1 checked by the caller)."""
    ld = Cin if ld is None else ld
    return (Cin % 32 == 0 and (Cout % 64 == 0 or (Cout % 32 == 0 and not HALO_V1 and HALO_PRENORM_MINPIX == 0))
            and ((Ho % 8 == 0 and Wo % 32 == 0) or (Ho % 16 == 0 and Wo % 16 == 0)) and ld % 8 == 0 and in_off % 8 == 0)


def conv(x, w, bias=None, *, stride=1, pad=1, ksize=3, down=False, upsample=False, pro=None, pro_act=L.PRO_NONE,
         act=L.ACT_NONE, residual=None, aux=None, aux_w=1.0, cin=None, in_off=0, out=None, split_k=None, wb=None,
         mma=None, stats=False, out_bf16=False):
    """x [N,H,W,ld] -> [N,Ho,Wo,Cout].  ``w`` packed [Cout,KH,KW,Cin].  ``cin``/``in_off`` select a channel
    slice of a wider input buffer.  ``down`` = VQGAN Downsample geometry (pad right/bottom only, stride 2)."""
    N, H, W, ld = x.shape
    Cout = w.shape[0]
    Cin = ld if cin is None else cin
    KH = KW = ksize
    assert w.numel() == Cout * KH * KW * Cin, (w.shape, Cout, KH, KW, Cin)
    Hv, Wv = (2 * H, 2 * W) if upsample else (H, W)
    if down:
        stride, pad_t, pad_l = 2, 0, 0
        Ho, Wo = Hv // 2, Wv // 2
    else:
        pad_t = pad_l = pad
        Ho = (Hv + 2 * pad - KH) // stride + 1
        Wo = (Wv + 2 * pad - KW) // stride + 1
    M = N * Ho * Wo
    mma = MMA if mma is None else mma
    out_bf16 = bool(out_bf16) and mma == L.MMA_BF16 and Cout % 4 == 0 and residual is None
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    in_dtype = L.BF16 if x.dtype == torch.bfloat16 else L.F32
    halo = (mma == L.MMA_BF16 and ksize == 3 and stride == 1 and not down and pad == 1
            and halo_bf16_eligible(Cin, Cout, Ho, Wo, ld, in_off)
            and (not out_bf16 or (Cout % 64 == 0 and not HALO_V1 and HALO_PRENORM_MINPIX == 0 and split_k in (None, 1))))
    assert in_dtype == L.F32 or halo, "bf16 activation tensors only feed the 3x3 halo convolution (via the normalise pass)"
    if halo and out_bf16:
        split_k = 1
    # fp32 policy: persistent LDS-halo kernel on f32 MFMA (16-channel chunks, GroupNorm affine + activation fused in staging)
    halo_f32 = (mma != L.MMA_BF16 and HALO_F32 and x.dtype == torch.float32 and ksize == 3 and stride == 1 and not down
                and pad == 1 and Cin % 16 == 0 and Cout % 32 == 0
                and ((Ho % 8 == 0 and Wo % 32 == 0) or (Ho % 16 == 0 and Wo % 16 == 0)) and ld % 4 == 0 and in_off % 4 == 0)
    if halo and (pro is not None or pro_act != L.PRO_NONE) and N * H * W >= HALO_PRENORM_MINPIX:
        # optional two-pass variant: normalise + activate once per element into a bf16 tensor in front of the halo conv
        # (default: the halo kernel applies the affine + activation itself while staging its fp32 halo)
        assert in_off == 0 and Cin == ld
        x16 = torch.empty((N, H, W, ld), dtype=torch.bfloat16, device=x.device)
        L.call('keep_norm_act_bf16', x, None if pro is None else pro[0], None if pro is None else pro[1], x16,
               N, H * W, ld, pro_act, in_dtype)
        x, pro, pro_act, in_dtype = x16, None, L.PRO_NONE, L.BF16
    # <= 4 output channels (the generator's 64 -> 3 output conv): exact-fp32 VALU kernel on an LDS halo, both policies
    cout4 = (COUT4 and Cout <= 4 and ksize == 3 and stride == 1 and not down and pad == 1 and not upsample
             and x.dtype == torch.float32 and not out_bf16 and Cin % 16 == 0 and ld % 4 == 0 and in_off % 4 == 0
             and Ho % 8 == 0 and Wo % 32 == 0 and residual is None and aux is None)
    # RGB first convolutions (Cin <= 3), bf16 policy: persistent im2col-in-LDS kernel
    c3 = (C3 and not cout4 and mma == L.MMA_BF16 and ksize == 3 and stride == 1 and not down and pad == 1 and not upsample
          and Cin <= 3 and Cout % 4 == 0 and Cout >= 32 and x.dtype == torch.float32 and not out_bf16 and Ho % 8 == 0
          and Wo % 32 == 0 and pro is None and pro_act == L.PRO_NONE and residual is None and aux is None
          and split_k in (None, 1))
    if c3:
        split_k = 1
    if cout4:
        split_k, halo, halo_f32, stats, mma = 1, False, False, False, L.MMA_F32
    elif mma == L.MMA_BF16 and wb is None:
        wb = bf16_twin(w)
    nsteps = KH * KW * math.ceil(Cin / (64 if mma == L.MMA_BF16 else 16))
    if mma != L.MMA_BF16 and Cin < 8 and pro is None and pro_act == L.PRO_NONE:      # f32 gather kernel: flattened K
        nsteps = math.ceil(KH * KW * Cin / 16)
    # latency-bound gather layers (few 64x64 output tiles, deep K): 256-channel K steps, single LDS buffer
    bk256 = (USE_BK256 and mma == L.MMA_BF16 and not halo and Cout > 32 and (Cout <= 64 or M <= 4096) and Cin >= 256 and nsteps >= 8)
    if bk256:
        nsteps = KH * KW * math.ceil(Cin / 256)
        if split_k is None and not out_bf16:
            tiles = math.ceil(M / 64) * math.ceil(Cout / 64)
            split_k = max(1, min(512 // max(tiles, 1), nsteps // 2, 16))
    if split_k is None:
        if out_bf16:
            split_k = 1
        elif halo_f32:  # 256-pixel x 64-channel work items on a persistent grid of 2 blocks per CU; split over 16-channel chunks
            items = (M // 256) * ((Cout + 63) // 64)
            split_k = 1 if items >= 256 else max(1, min(512 // items, Cin // 32, 16))
        elif halo:     # 8x32-pixel x 64-channel tiles; split over the 32-channel Cin chunks
            waves = (M // 256) * ((Cout + 63) // 64) * 4
            split_k = 1 if waves >= _TARGET_WAVES else max(1, min(_TARGET_WAVES // waves, Cin // 64, 16))
        else:
            split_k = pick_split_k(M, Cout, nsteps, mma == L.MMA_BF16)
    ws = empty((split_k * M * Cout,), x) if split_k > 1 else None
    # per-tile channel statistics of the output for the next GroupNorm / InstanceNorm (epilogue-fused)
    part, stats_P = None, 0
    if stats and split_k == 1:
        halo_v2 = halo_f32 or (halo and pro is None and pro_act == L.PRO_NONE and not HALO_V1)
        bm = 64 if c3 else (64 if halo_v2 else 256) if (halo or halo_f32) else (128 if Cout <= 32 else (64 if (Cout <= 64 or M <= 4096) else 128))
        if (Ho * Wo) % bm == 0 and out.shape[-1] == Cout:
            stats_P = (Ho * Wo) // bm
            part = empty((N, stats_P, Cout, 2), x)
    xin = x if in_off == 0 else x.view(-1)[in_off:]
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tw = 32 if (Ho % 8 == 0 and Wo % 32 == 0) else 16
        kname = ('conv3x3_cout4_kernel' if cout4 else 'conv3x3_c3_kernel' if c3 else f'conv3x3_halo_f32_kernel<{tw}>' if halo_f32 else
                 f"conv3x3_halo3_kernel<{'true' if in_dtype == L.BF16 else 'false'}, {tw}>" if halo else
                 tile_config(M, Cout, mma == L.MMA_BF16, bk256,
                             plain=(Cin % 8 == 0 and ld % 4 == 0 and pro is None and pro_act == L.PRO_NONE and not upsample)))
        # algorithmic bytes: one read of the input window at its storage type, the weights, one write of the output
        alg_bytes = (N * H * W * Cin * x.element_size() + Cout * KH * KW * Cin * (2 if mma == L.MMA_BF16 else 4)
                     + M * Cout * out.element_size() + (0 if residual is None else M * Cout * 4))
        PROFILE.append((kname, 2.0 * M * Cout * KH * KW * Cin, split_k, e0, e1, alg_bytes))
        e0.record()
    L.conv2d(inp=xin, weight=w, bias=bias, out=out, pro_scale=None if pro is None else pro[0],
             pro_shift=None if pro is None else pro[1], residual=residual, aux=aux, workspace=ws,
             N=N, H=H, W=W, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride=stride, pad_t=pad_t, pad_l=pad_l, Ho=Ho, Wo=Wo,
             in_ld=ld, out_ld=out.shape[-1], res_ld=0 if residual is None else residual.shape[-1],
             upsample=int(upsample), pro_act=pro_act, epi_act=act, aux_w=float(aux_w), split_k=split_k, dtype=in_dtype,
             mma=mma, weight_bf16=wb if mma == L.MMA_BF16 else None, stats_out=part, stats_P=stats_P, bk256=int(bk256),
             out_dtype=L.BF16 if out_bf16 else L.F32)
    if part is not None:
        out._keep_stats = (part, stats_P)
    if PROFILE is not None:
        PROFILE[-1][4].record()
    return out


def linear(x, w, bias=None, *, act=L.ACT_NONE, residual=None, pro=None, pro_act=L.PRO_NONE, cin=None, in_off=0,
           n_img=1, out_bf16=False):
    """x [M,ld] (or any [...,ld]) @ w[Cout,Cin]^T.  ``n_img``>1 makes the prologue per-image: rows are n_img
    images of M/n_img pixels each (1x1 conv on a feature map)."""
    shp = x.shape
    ld = shp[-1]
    M = x.numel() // ld
    if (TOKEN_LINEAR and MMA == L.MMA_BF16 and ld == 128 and w.shape[0] in (128, 256, 384) and w.shape[-1] == 128 and M >= 65536
            and act == L.ACT_NONE and residual is None and pro is None and pro_act == L.PRO_NONE and cin is None and in_off == 0
            and x.dtype == torch.float32 and x.is_contiguous()):
        return token_linear(x.view(M, ld), w.view(w.shape[0], 128), bias, out_bf16).reshape(*shp[:-1], w.shape[0])
    x4 = x.reshape(n_img, M // n_img, 1, ld)
    res4 = None if residual is None else residual.reshape(n_img, M // n_img, 1, residual.shape[-1])
    y = conv(x4, w, bias, stride=1, pad=0, ksize=1, pro=pro, pro_act=pro_act, act=act, residual=res4, cin=cin,
             in_off=in_off, out_bf16=out_bf16)
    return y.reshape(*shp[:-1], w.shape[0])


def norm_affine(x, gamma, beta, groups, eps):
    """GroupNorm / InstanceNorm statistics of x [N,H,W,C] -> (scale, shift) [N,C] for a conv prologue.
    groups == C and gamma=None -> InstanceNorm2d(affine=False)."""
    N, H, W, C = x.shape
    HW = H * W
    scale = empty((N, C), x)
    shift = empty((N, C), x)
    fused = getattr(x, '_keep_stats', None)
    if fused is not None:      # the producing conv already reduced (sum, sumsq) per tile in its epilogue
        part, P = fused
        L.call('keep_norm_finalize', part, gamma, beta, scale, shift, N, HW, C, groups, P, float(eps))
        return scale, shift
    assert x.dtype == torch.float32, "bf16 activations carry their statistics from the producing conv's epilogue"
    cpg = C // groups
    if ((cpg % 4 == 0 and HW * cpg <= 32768) or HW <= 1024) and C % 4 == 0 and groups * N >= 16:
        # small maps: one block per (image, group), one launch
        L.call('keep_group_stats', x, gamma, beta, scale, shift, N, HW, C, groups, float(eps))
        return scale, shift
    P = max(1, min(HW // 64, 1024))
    part = empty((N, P, C, 2), x)
    L.call('keep_chan_stats', x, part, N, HW, C, C, P)
    L.call('keep_norm_finalize', part, gamma, beta, scale, shift, N, HW, C, groups, P, float(eps))
    return scale, shift


def layernorm(x, gamma, beta, *, res=None, pos=None, eps=1e-5):
    """LN over the last dim.  Returns y (+res); with ``pos`` ([P,C], broadcast over rows mod P) also y+pos."""
    C = x.shape[-1]
    M = x.numel() // C
    out = torch.empty_like(x)
    out2 = torch.empty_like(x) if pos is not None else None
    L.call('keep_layernorm', x, gamma, beta, res, out, pos, 0 if pos is None else pos.shape[0], out2, M, C, float(eps))
    return out if pos is None else (out, out2)


def geglu(x):
    F = x.shape[-1] // 2
    M = x.numel() // (2 * F)
    out = empty((*x.shape[:-1], F), x)
    L.call('keep_geglu', x, out, M, F)
    return out


def attention(q, k, v, o, *, B, H, Lq, Lk, D, Dv, scale, q_str, k_str, v_str, o_str, mode=0, T=0, seg_len=0,
              img_h=0, img_w=0, ksplit=0, shift=0, kv_rot=0, n_img=0, mma=None):
    """Strides are (batch, token, head) element strides."""
    mma = ATTN_MMA if mma is None else mma
    in_dtype = L.F32
    if q.dtype == torch.bfloat16:
        assert k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16
        in_dtype, mma = L.BF16, L.MMA_BF16
    elif mma == L.MMA_BF16 and (D % 16 or any(v % 4 for v in (*q_str, *k_str))):
        mma = L.MMA_F32
    L.attention(q=q, k=k, v=v, o=o,
                q_bs=q_str[0], q_ts=q_str[1], q_hs=q_str[2], k_bs=k_str[0], k_ts=k_str[1], k_hs=k_str[2],
                v_bs=v_str[0], v_ts=v_str[1], v_hs=v_str[2], o_bs=o_str[0], o_ts=o_str[1], o_hs=o_str[2],
                B=B, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=float(scale), mode=mode, T=T, seg_len=seg_len,
                img_h=img_h, img_w=img_w, ksplit=ksplit, shift=shift, kv_rot=kv_rot, n_img=n_img, mma=mma, in_dtype=in_dtype)
    return o


def token_linear(x, w, bias=None, out_bf16=False):
    """Streaming GEMM for the GMFlow projections (bf16 policy): x [M,128] fp32 @ w[N,128]^T, N in {128,256,384}."""
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    L.call('keep_token_linear', x, bf16_twin(w), bias, out, M, K, N, L.BF16 if out_bf16 else L.F32)
    return out


def gm_mlp(a, b, w0, w2):
    """GMFlow FFN fused (bf16 policy): W2 . gelu(W0 . cat[a | b]); a, b [M,C] fp32, w0 [8C,2C], w2 [C,8C] fp32 views of
    the packed blob (their bf16 twins are used)."""
    C = a.shape[-1]
    M = a.numel() // C
    out = empty((M, C), a)
    L.call('keep_gm_mlp', a, b, bf16_twin(w0), bf16_twin(w2), out, M, C)
    return out


def offset(t, off):
    """Flat view of ``t`` starting ``off`` elements in (channel-slice pointer for strided kernels)."""
    return t.view(-1)[off:] if off else t


def concat2(a, b):
    C1, C2 = a.shape[-1], b.shape[-1]
    M = a.numel() // C1
    out = empty((*a.shape[:-1], C1 + C2), a)
    L.call('keep_concat2', a, b, out, M, C1, C2)
    return out


def add_bcast(a, t, alpha=1.0):
    out = torch.empty_like(a)
    L.call('keep_add_bcast', a, t, out, a.numel(), t.numel(), float(alpha))
    return out


def nchw_to_nhwc(x, mode=0):
    N, C, H, W = x.shape
    out = empty((N, H, W, C), x)
    L.call('keep_nchw_to_nhwc', x, out, N, C, H * W, mode)
    return out


def nhwc_to_nchw(x):
    N, H, W, C = x.shape
    out = empty((N, C, H, W), x)
    L.call('keep_nhwc_to_nchw', x, out, N, C, H * W)
    return out
