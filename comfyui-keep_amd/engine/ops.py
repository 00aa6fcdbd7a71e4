"""Tensor-level wrappers over the C-ABI: shape bookkeeping + output allocation only.

All activations are channels-last fp32 device tensors: feature maps ``[N,H,W,C]``, token
matrices ``[M,C]`` (same memory).  torch is used for allocation (caching allocator -> safe
inside hipGraph capture) and views; every arithmetic op is a ``libkeep_hip.so`` kernel.

Nothing here knows how ``keep_conv2d`` tiles its work: kernel choice, split-K, workspace and
statistics sizes come from ``keep_conv2d_plan`` (cached per argument signature).  The precision
policy and the weight blobs it needs live on an ``Ops`` INSTANCE owned by one ``KeepNet`` -- two
nets (or two threads) with different policies never share state.  The module-level functions
are the methods of a default fp32 instance (kernel tests call them with explicit ``mma=``).
"""
import os

import torch

from . import hiplib as L

TOKEN_LINEAR = os.environ.get('KEEP_NO_TOKEN_LINEAR') is None   # dev switch: streaming GEMM for the GMFlow projections
UP2_PHASES = os.environ.get('KEEP_X3_UP2', '1') != '0'      # x3 policy: nearest x2 + 3x3 as four 2x2-tap phase convolutions (A/B: 0)
FUSE_LN = os.environ.get('KEEP_X3_FUSE_LN', '1') != '0'      # x3 policy: GMFlow `merge -> norm1` / `mlp.2 -> norm2` LayerNorms in the GEMM epilogue (A/B: 0)
USE_BK256 = bool(int(os.environ.get('KEEP_BK256', '0')))   # measured slower than BK=64 + split-K at B<=4 (kept for A/B)
# bf16 policy: inputs of fewer pixels (N*H*W) than this skip the normalise+activate -> bf16 pass in front of the halo
# conv and use the kernel variant that applies the prologue while staging (A/B switch, default: always the two-pass form)
HALO_PRENORM_MINPIX = int(os.environ.get('KEEP_HALO_PRENORM_MINPIX', '0'))

DEBUG_SYNC = os.environ.get('KEEP_DEBUG_SYNC') is not None
_PLAN_CACHE = {}
# Deployment settings of the library, read ONCE here (the library itself reads no environment variable: they travel in the argument
# structs).  KEEP_PLAN_REF_IMAGES: the fixed reference batch of the parity policies' plans (default 16; 2 = latency profile for single
# clips; results are batch-invariant within one value).  KEEP_X3_EXACT_ACT=1: library expf / erff inside the x3 kernels.
PLAN_REF_IMAGES = int(os.environ.get('KEEP_PLAN_REF_IMAGES', '0') or 0)
DEFAULT_CONV_FLAGS = L.CONV_X3_EXACT_ACT if os.environ.get('KEEP_X3_EXACT_ACT') else 0


class Plan:
    __slots__ = ('split_k', 'ws_floats', 'stats_P', 'wants_bf16_input', 'out_bf16_ok', 'out_amax_ok', 'kernel')

    def __init__(self, o):
        self.split_k = o.split_k
        self.ws_floats = o.workspace_bytes // 4
        self.stats_P = o.stats_P
        self.wants_bf16_input = bool(o.wants_bf16_input)
        self.out_bf16_ok = bool(o.out_bf16_ok)
        self.out_amax_ok = bool(o.out_amax_ok)
        self.kernel = o.kernel.decode()


class Stats:
    """What a producing convolution knows about its output for free (reduced in its epilogue): per-channel (sum, sumsq)
    partials [N,P,C,2] for the next GroupNorm / InstanceNorm, and (x3 policy) the per-image max |y| [N] -- the range probe
    of the next x3 operator that reads the tensor un-normalised.  Either may be None."""
    __slots__ = ('part', 'P', 'amax')

    def __init__(self, part=None, P=0, amax=None):
        self.part, self.P, self.amax = part, P, amax


def _plan(a, key):
    """keep_conv2d_plan for these arguments, cached by everything the decision can depend on (shapes, flags, which
    optional tensors exist, pointer alignment class) -- never by values."""
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        pl = _PLAN_CACHE[key] = Plan(L.conv2d_plan(a))
    return pl


def empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def offset(t, off):
    """Flat view of ``t`` starting ``off`` elements in (channel-slice pointer for strided kernels)."""
    return t.view(-1)[off:] if off else t


class Ops:
    def __init__(self):
        # matrix-core operand precision of keep_conv2d / keep_attention launches:
        #   L.MMA_F32 exact f32 (parity), L.MMA_X3 split fp16 x 3 (parity-grade fast policy), L.MMA_BF16 (speed policy)
        self.mma = L.MMA_F32
        self.attn_mma = L.MMA_F32
        self.blob32 = None      # packed fp32 weight blob the twins below are resolved against (same element offsets)
        self.blob16 = None      # bf16 twin
        self.blobx3 = None      # split-fp16 twin: int16 tensor, 2 elements per weight, per-tensor [.., Cin/16, hi16|lo16]
        self.x3_acc_scale = 1.0
        self._up2, self._up2_src = {}, (None, None)
        # bench.py's roofline leg: when a list, every keep_conv2d launch is bracketed by HIP events on the launch stream
        # and appended as (kernel family, algorithmic_flops, split_k, start_event, end_event, algorithmic_bytes)
        self.profile = None
        self.amax_arena = None     # x3: per-forward arena of fused max|out| slots (begin_forward)
        self.amax_pos = 0
        self.status = None         # one device int32: KEEP_STATUS_* bits raised by this forward's kernels (begin_forward zeroes it)
        self.arena_generation = 0  # bumped when the block is (re)allocated: hipGraphs captured before that are stale
        # keep_conv2d_args.flags / .plan_ref_images and keep_attention_args.flags of every launch of this Ops (kernel-selection
        # overrides for tests and A/B runs: hiplib.CONV_* / ATTN_*; the plan's reference batch: a per-net numerics setting)
        self.flags = DEFAULT_CONV_FLAGS
        self.attn_flags = 0
        self.plan_ref_images = PLAN_REF_IMAGES

    def begin_forward(self, device):
        """Zero this forward's bookkeeping words with ONE fill launch: the status word (non-finite logits / tensors, see
        keep_argmax_gather, keep_nonfinite_flag) and, for the x3 policy, the arena of per-image max|out| slots the convolutions
        fill with atomicMax (instead of one zero-fill launch per convolution: 2.4 k launches per 16-clip step).  A slot lives
        until the next begin_forward: Stats objects never outlive the forward pass that made them.
        The block is allocated ONCE per Ops and device and never freed or swapped while the Ops lives: captured hipGraphs
        bake its address in, so dropping it on a policy change (as round 2 did) left replays writing through a dangling pointer."""
        self.ensure_arena(device)
        self._block.zero_()
        self.amax_pos = 0

    def ensure_arena(self, device):
        device = torch.device(device)
        if self.amax_arena is None or self.amax_arena.device != device:
            self._block = torch.empty((1 << 18) + 4, dtype=torch.float32, device=device)
            self.amax_arena = self._block[4:]
            self.status = self._block[:1].view(torch.int32)
            self.arena_generation += 1

    def set_precision(self, mma, blob32=None, blob16=None, blobx3=None, x3_acc_scale=1.0, x3_scales=None):
        """x3_scales: per-tensor accumulator scales of the split-fp16 twin, [(first element, one past the last, 2^-e), ...] sorted
        by offset (``make_x3_blob``); without it every tensor of the blob carries ``x3_acc_scale``."""
        self._x3_table = None
        if x3_scales:
            self._x3_table = ([a for a, _, _ in x3_scales], list(x3_scales))
        self.mma = self.attn_mma = mma
        if blobx3 is not None and (blob32 is not self._up2_src[0] or blobx3 is not self._up2_src[1]):
            # phase kernels of the Upsample convolutions (up2_twin): derived from THESE blob objects -- a new upload, even one that lands
            # on the same addresses, starts from an empty cache; a policy switch on the same blobs keeps it (captured x3 graphs hold
            # the addresses of its tensors)
            self._up2, self._up2_src = {}, (blob32, blobx3)
        self.blob32, self.blob16, self.blobx3, self.x3_acc_scale = blob32, blob16, blobx3, float(x3_acc_scale)

    # ------------------------------------------------------------------ weight twins
    def _blob_off(self, w):
        if self.blob32 is None:
            raise RuntimeError("this precision policy needs the packed weight blobs registered (Ops.set_precision)")
        off = (w.data_ptr() - self.blob32.data_ptr()) // 4
        if off < 0 or off + w.numel() > self.blob32.numel() or not w.is_contiguous():
            raise RuntimeError("weight view is not inside the packed blob; pass the twin explicitly")
        return off

    def bf16_twin(self, w):
        """bf16 copy of an fp32 weight view that lives inside the registered packed blob."""
        if self.blob16 is None:
            raise RuntimeError("bf16 policy needs the bf16 blob registered (Ops.set_precision)")
        off = self._blob_off(w)
        return self.blob16[off:off + w.numel()]

    def up2_twin(self, w):
        """(weight_x3, acc_scale) of the four 2x2-tap phase kernels of ``w`` (packed [Cout,3,3,Cin]) for
        ``upsample = UPSAMPLE_X2_PHASES`` -- built once per weight tensor (the first, eager, call) and kept for the lifetime of the
        blob it points into."""
        key = (w.data_ptr(), tuple(w.shape))
        tw = self._up2.get(key)
        if tw is None:
            w4 = up2_phase_weights(w)
            sc = x3_scale_for(float(w4.abs().max()))
            tw = self._up2[key] = (split_x3(w4.reshape(-1, w.shape[-1]), sc).view(-1), 1.0 / sc)
        return tw

    def ffn_w2_twin(self, w2):
        """(weight_x3, acc_scale) of ``w2`` [C, hidden] for ``keep_gm_ffn_x3``: every group of 16 hidden units in the order
        [0-3, 8-11, 4-7, 12-15] (``ffn_w2_perm``), split with the tensor's own power-of-two scale.  Built once per weight tensor
        (outside any stream capture: ``KeepNet._activate_precision``) and kept with the up2 twins for the lifetime of the blob."""
        key = ('ffn_w2', w2.data_ptr(), tuple(w2.shape))
        tw = self._up2.get(key)
        if tw is None:
            wp = ffn_w2_perm(w2)
            sc = x3_scale_for(float(wp.abs().max()))
            tw = self._up2[key] = (split_x3(wp, sc).view(-1), 1.0 / sc)
        return tw

    def gm_ffn_x3(self, src, msg, w0, w2, gamma, beta, eps):
        """GMFlow FFN + norm2 + residual as one launch (x3 policy): LayerNorm(W2 . gelu(W0 . cat[src | msg])) + src; src, msg [M, C]
        fp32, w0 [8C, 2C] / w2 [C, 8C] fp32 views of the packed blob (their x3 twins are used)."""
        C = src.shape[-1]
        M = src.numel() // C
        out = empty((M, C), src)
        w2p, asc2 = self.ffn_w2_twin(w2)
        if self.profile is not None:            # bench.py's roofline leg: the fused block counts with the conv / GEMM path it replaces
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            Hd = w0.shape[0]
            self.profile.append(('gm_ffn_x3_kernel', 2.0 * M * (2 * C * Hd + Hd * C), 1, e0, e1,
                                 3 * M * C * 4 + (2 * C * Hd + Hd * C) * 4, (M // 4096 if M % 4096 == 0 else 1, M, 1, 2 * C, C, 1, 1, 0, False)))
            e0.record()
        L.call('keep_gm_ffn_x3', src, msg, self.x3_twin(w0), float(self.x3_scale_of(w0)), w2p, float(asc2), gamma, beta, float(eps), out,
               M, C, w0.shape[0], 1 if (self.flags & L.CONV_X3_EXACT_ACT) else 0)
        if self.profile is not None:
            self.profile[-1][4].record()
        return out.view(src.shape)

    def x3_twin(self, w):
        """split-fp16 copy of an fp32 weight view (row slices of a [Cout, .., Cin] tensor keep their layout), or None
        when the policy has no x3 blob / the tensor's Cin is not a multiple of 16 (such layers run on the f32 kernels)."""
        if self.blobx3 is None or w.shape[-1] % 16:
            return None
        off = self._blob_off(w)
        return self.blobx3[2 * off:2 * (off + w.numel())]

    def x3_scale_of(self, w):
        """Accumulator scale (2^-e) of the twin of weight view ``w``: its own tensor's (a row slice shares its tensor's scale)."""
        if getattr(self, '_x3_table', None) is None:
            return self.x3_acc_scale
        import bisect
        off = self._blob_off(w)
        starts, rows = self._x3_table
        i = bisect.bisect_right(starts, off) - 1
        if i < 0 or not (rows[i][0] <= off and off + w.numel() <= rows[i][1]):
            raise RuntimeError("weight view does not lie inside one tensor of the x3 blob")
        return rows[i][2]

    # ------------------------------------------------------------------ keep_conv2d
    def conv(self, x, w, bias=None, *, stride=1, pad=1, ksize=3, down=False, upsample=False, pro=None, pro_act=L.PRO_NONE,
             act=L.ACT_NONE, residual=None, aux=None, aux_w=1.0, cin=None, in_off=0, out=None, split_k=None, wb=None,
             wx3=None, x3_acc_scale=None, mma=None, stats=False, out_bf16=False, bounded=False, x_amax=None, x2=None,
             reflect=False, out_ld=None, ln=None, out_hw=None):
        """x [N,H,W,ld] -> [N,Ho,Wo,Cout].  ``w`` packed [Cout,KH,KW,Cin].  ``cin``/``in_off`` select a channel
        slice of a wider input buffer.  ``down`` = VQGAN Downsample geometry (pad right/bottom only, stride 2).
        ``stats=True`` returns ``(out, st)``: ``st`` is a ``Stats`` (epilogue-reduced GroupNorm partials and, under the x3
        policy, the per-image max |out|), or None when this launch could emit neither (split-K).  ``x_amax``: the
        producer's ``Stats.amax`` of x (any upper bound of max |x| per image works), replacing the range probe.
        ``bounded=True``: the caller vouches that |x| stays far below the fp16 range (normalised / attention-averaged
        inputs); otherwise an x3 launch without a normalising prologue first probes the input range (keep_absmax).
        ``ln=(gamma, beta, eps)``: LayerNorm over the output channels in the epilogue, BEFORE the residual is added (x3 GEMM form,
        Cout == 128, rows % 128 == 0: ``ln_fusable``); the library refuses anything else."""
        N, H, W, ld = x.shape
        Cout = w.shape[0]
        Cin = ld if cin is None else cin
        if x2 is not None:          # K-concatenated second input (x3 GEMM form, keep_conv2d in2): channels ld.. come from x2
            assert bounded and pro is None and cin is None and in_off == 0, 'x2 needs a bounded, unsliced, prologue-free input'
            Cin = ld + x2.shape[-1]
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
        if out_hw is not None:      # fewer output rows / columns than the symmetric padding gives (an even kernel: pad only top / left)
            assert out_hw[0] <= Ho and out_hw[1] <= Wo
            Ho, Wo = out_hw
        M = N * Ho * Wo
        mma = self.mma if mma is None else mma
        in_dtype = L.BF16 if x.dtype == torch.bfloat16 else L.F32
        if mma == L.MMA_BF16 and wb is None and self.blob16 is not None:
            wb = self.bf16_twin(w)      # (without a twin the library still accepts the exact-fp32 Cout <= 4 kernel)
        up_mode = L.UPSAMPLE_X2_PHASES if upsample == L.UPSAMPLE_X2_PHASES and upsample is not True else int(bool(upsample))   # (explicit 2: the caller brings phase weights)
        if (upsample and mma == L.MMA_X3 and wx3 is None and UP2_PHASES and KH == 3 and stride == 1 and pad == 1 and not down
                and pro is None and pro_act == L.PRO_NONE and act == L.ACT_NONE and aux is None and x2 is None and not reflect
                and split_k in (None, 1) and H % 8 == 0 and W % 32 == 0 and Cin % 16 == 0 and Cout % 64 == 0 and in_dtype == L.F32
                and not out_bf16 and self.x3_twin(w) is not None):
            # nearest x2 + 3x3 as four 2x2-tap phase convolutions on the source grid: 4 of 9 taps are multiplied
            wx3, x3_acc_scale = self.up2_twin(w)
            up_mode = L.UPSAMPLE_X2_PHASES
        if mma == L.MMA_X3 and wx3 is None:
            wx3 = self.x3_twin(w)
            if wx3 is not None and x3_acc_scale is None:
                x3_acc_scale = self.x3_scale_of(w)
        if x3_acc_scale is None:
            x3_acc_scale = self.x3_acc_scale
        want_bf16_out = bool(out_bf16) and mma == L.MMA_BF16 and residual is None
        xin = x if in_off == 0 else x.view(-1)[in_off:]
        in_amax = None
        if mma == L.MMA_X3 and (wx3 is not None or (Cin <= 3 and KH == 3)) and pro is None and not bounded:   # (RGB convs split fp32 weights in-kernel)
            in_amax = x_amax if (x_amax is not None and x_amax.numel() == N) else absmax(xin, N, H * W, Cin, ld, H * W * ld, self)
        out_ld = (Cout if out is None else out.shape[-1]) if out_ld is None else out_ld

        def make_args(inp, dtype, pro_t, pro_a, odt, sk):
            return L.conv_args(
                inp=inp, weight=w, bias=bias, out=out, pro_scale=None if pro_t is None else pro_t[0],
                pro_shift=None if pro_t is None else pro_t[1], residual=residual, aux=aux, workspace=None,
                N=N, H=H, W=W, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride=stride, pad_t=pad_t, pad_l=pad_l, Ho=Ho, Wo=Wo,
                in_ld=ld, out_ld=out_ld, res_ld=0 if residual is None else residual.shape[-1],
                upsample=up_mode, pro_act=pro_a, epi_act=act, aux_w=float(aux_w), split_k=sk, dtype=dtype,
                mma=mma, weight_bf16=wb if mma == L.MMA_BF16 else None, stats_out=None, stats_P=0,
                bk256=int(USE_BK256), out_dtype=odt, weight_x3=wx3 if mma == L.MMA_X3 else None,
                x3_acc_scale=float(x3_acc_scale), x3_in_amax=in_amax, x3_out_amax=None,
                in2=x2, in2_cin1=0 if x2 is None else ld, pad_mode=L.PAD_REFLECT if reflect else L.PAD_ZERO,
                ln_gamma=None if ln is None else ln[0], ln_beta=None if ln is None else ln[1],
                ln_eps=0.0 if ln is None else float(ln[2]), flags=self.flags, plan_ref_images=self.plan_ref_images)

        def key_of(dtype, pro_t, pro_a, odt, sk):
            return (N, H, W, ld, Cin, Cout, KH, stride, pad_t, pad_l, Ho, Wo, out_ld, up_mode, pro_a, act, dtype, mma,
                    odt, sk, pro_t is not None, residual is not None, 0 if residual is None else residual.shape[-1],
                    aux is not None, bias is not None, in_off % 8, wx3 is not None, USE_BK256, x2 is not None, bool(reflect),
                    ln is not None, self.flags, self.plan_ref_images)

        sk_req = 0 if split_k is None else int(split_k)
        odt = L.BF16 if want_bf16_out else L.F32
        a = make_args(xin, in_dtype, pro, pro_act, odt, sk_req)
        pl = _plan(a, key_of(in_dtype, pro, pro_act, odt, sk_req))
        if (pl.wants_bf16_input and in_off == 0 and Cin == ld and N * H * W >= HALO_PRENORM_MINPIX):
            # bf16 policy, 3x3 halo geometry with a prologue: normalise + activate once per element into a bf16 tensor
            x16 = torch.empty((N, H, W, ld), dtype=torch.bfloat16, device=x.device)
            L.call('keep_norm_act_bf16', x, None if pro is None else pro[0], None if pro is None else pro[1], x16,
                   N, H * W, ld, pro_act, in_dtype)
            x = xin = x16
            pro, pro_act, in_dtype = None, L.PRO_NONE, L.BF16
            a = make_args(xin, in_dtype, None, pro_act, odt, sk_req)
            pl = _plan(a, key_of(in_dtype, None, pro_act, odt, sk_req))
        if want_bf16_out and not pl.out_bf16_ok:
            want_bf16_out, odt = False, L.F32
            a = make_args(xin, in_dtype, pro, pro_act, odt, sk_req)
            pl = _plan(a, key_of(in_dtype, pro, pro_act, odt, sk_req))
        if out is None:
            out = torch.empty((N, Ho, Wo, Cout), dtype=torch.bfloat16 if want_bf16_out else torch.float32, device=x.device)
            a.out = out.data_ptr()
        a.split_k = pl.split_k
        ws = empty((pl.ws_floats,), x) if pl.ws_floats else None
        if ws is not None:
            a.workspace = ws.data_ptr()
        st = None
        if stats and ((pl.stats_P and stats != 'amax') or pl.out_amax_ok):
            st = Stats()
            if pl.stats_P and stats != 'amax':      # ('amax': only the fused max|out| -- a GEMM whose consumer needs a range, not a GroupNorm)
                st.part, st.P = empty((N, pl.stats_P, Cout, 2), x), pl.stats_P
                a.stats_out, a.stats_P = st.part.data_ptr(), pl.stats_P
            if pl.out_amax_ok:
                if self.amax_arena is not None and self.amax_pos + N <= self.amax_arena.numel():
                    st.amax = self.amax_arena[self.amax_pos:self.amax_pos + N]       # zeroed once per forward (begin_forward)
                    self.amax_pos += N
                    a.x3_out_amax_zeroed = 1
                else:
                    st.amax = empty((N,), x)
                a.x3_out_amax = st.amax.data_ptr()
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # algorithmic bytes: one read of the input window at its storage type, the weights, one write of the output
            alg_bytes = (N * H * W * Cin * x.element_size() + Cout * KH * KW * Cin * (2 if mma == L.MMA_BF16 else 4)
                         + M * Cout * out.element_size() + (0 if residual is None else M * Cout * 4))
            self.profile.append((pl.kernel, 2.0 * M * Cout * KH * KW * Cin, pl.split_k, e0, e1, alg_bytes,
                                 (N, H, W, Cin, Cout, KH, stride, int(upsample), pro is not None)))
            e0.record()
        if DEBUG_SYNC:      # dev aid: name every launch and wait for it, so a GPU fault is attributed to its kernel
            import sys
            print(f'[keep] {pl.kernel} N={N} H={H} W={W} ld={ld} Cin={Cin} Cout={Cout} k={KH} s={stride} up={int(upsample)} '
                  f'split={pl.split_k} in_off={in_off} pro={pro is not None}/{pro_act} act={act} res={residual is not None} '
                  f'aux={aux is not None} statsP={pl.stats_P if stats else 0}', file=sys.stderr, flush=True)
        L.conv2d_launch(a)
        if DEBUG_SYNC:
            torch.cuda.synchronize()
            if not bool(torch.isfinite(out.float()).all()):
                import sys
                print(f'[keep] NON-FINITE output of {pl.kernel}; input finite={bool(torch.isfinite(x.float()).all())} '
                      f'absmax={float(x.float().abs().max()):.4g} pro={None if pro is None else [float(t.abs().max()) for t in pro]} '
                      f'res absmax={None if residual is None else float(residual.abs().max())}', file=sys.stderr, flush=True)
        if self.profile is not None:
            self.profile[-1][4].record()
        return (out, st) if stats else out

    def linear(self, x, w, bias=None, *, act=L.ACT_NONE, residual=None, pro=None, pro_act=L.PRO_NONE, cin=None, in_off=0,
               n_img=None, out_bf16=False, bounded=False, x_amax=None, x2=None, ln=None, want_amax=False):
        """x [M,ld] (or any [...,ld]) @ w[Cout,Cin]^T.  ``n_img``: the rows are n_img independent images (frames, clips) of
        M/n_img rows each -- the unit of the per-image prologue AND of the library's plan (kernel tile / split-K are chosen
        from the per-image row count, so a clip's result never depends on its batch-mates).  Default: the leading axis of a
        3-D+ input, 1 for a plain [M,ld] matrix (callers that flatten independent images into rows pass it)."""
        shp = x.shape
        ld = shp[-1]
        M = x.numel() // ld
        if n_img is None:
            n_img = shp[0] if x.dim() >= 3 else 1
        assert M % n_img == 0, (tuple(shp), n_img)
        if (TOKEN_LINEAR and self.mma == L.MMA_BF16 and ld == 128 and w.shape[0] in (128, 256, 384) and w.shape[-1] == 128
                and M >= 65536 and act == L.ACT_NONE and residual is None and pro is None and pro_act == L.PRO_NONE
                and cin is None and in_off == 0 and x.dtype == torch.float32 and x.is_contiguous()):
            return self.token_linear(x.view(M, ld), w.view(w.shape[0], 128), bias, out_bf16).reshape(*shp[:-1], w.shape[0])
        x4 = x.reshape(n_img, M // n_img, 1, ld)
        x24 = None if x2 is None else x2.reshape(n_img, M // n_img, 1, x2.shape[-1])
        res4 = None if residual is None else residual.reshape(n_img, M // n_img, 1, residual.shape[-1])
        y = self.conv(x4, w, bias, stride=1, pad=0, ksize=1, pro=pro, pro_act=pro_act, act=act, residual=res4, cin=cin,
                      in_off=in_off, out_bf16=out_bf16, bounded=bounded, x_amax=x_amax, x2=x24, ln=ln, stats='amax' if want_amax else False)
        if want_amax:      # (y, per-image max |y| from the GEMM's epilogue, or None where the launch cannot emit it): the range probe of the consumer
            y, st = y
            return y.reshape(*shp[:-1], w.shape[0]), (None if st is None else st.amax)
        return y.reshape(*shp[:-1], w.shape[0])

    def ln_fusable(self, w, rows_per_image):
        """Can ``linear(x, w, ln=...)`` run its LayerNorm in the GEMM's epilogue?  (x3 policy with a split twin of ``w``, 128 output
        channels, K a multiple of 32, whole 128-row tiles per image -- a rule that never looks at the batch.)"""
        return (FUSE_LN and self.mma == L.MMA_X3 and w.shape[0] == 128 and w.shape[-1] % 32 == 0 and rows_per_image % 128 == 0
                and self.x3_twin(w) is not None)

    # ------------------------------------------------------------------ normalisation
    @staticmethod
    def norm_affine(x, gamma, beta, groups, eps, stats=None):
        """GroupNorm / InstanceNorm statistics of x [N,H,W,C] -> (scale, shift) [N,C] for a conv prologue.
        groups == C and gamma=None -> InstanceNorm2d(affine=False).  ``stats``: the (partials, P) the producing
        convolution reduced in its epilogue (saves one full read of x)."""
        N, H, W, C = x.shape
        HW = H * W
        scale = empty((N, C), x)
        shift = empty((N, C), x)
        if stats is not None and stats.part is not None:
            L.call('keep_norm_finalize', stats.part, gamma, beta, scale, shift, N, HW, C, groups, stats.P, float(eps))
            return scale, shift
        assert x.dtype == torch.float32, "bf16 activations carry their statistics from the producing conv's epilogue"
        cpg = C // groups
        if ((cpg % 4 == 0 and HW * cpg <= 32768) or HW <= 1024) and C % 4 == 0 and groups >= 2:   # (per-image rule: never N)
            # small maps: one block per (image, group), one launch
            L.call('keep_group_stats', x, gamma, beta, scale, shift, N, HW, C, groups, float(eps))
            return scale, shift
        P = max(1, min(HW // 64, 1024))
        part = empty((N, P, C, 2), x)
        L.call('keep_chan_stats', x, part, N, HW, C, C, P)
        L.call('keep_norm_finalize', part, gamma, beta, scale, shift, N, HW, C, groups, P, float(eps))
        return scale, shift

    @staticmethod
    def layernorm(x, gamma, beta, *, res=None, pos=None, eps=1e-5):
        """LN over the last dim.  Returns y (+res); with ``pos`` ([P,C], broadcast over rows mod P) also y+pos."""
        C = x.shape[-1]
        M = x.numel() // C
        out = torch.empty_like(x)
        out2 = torch.empty_like(x) if pos is not None else None
        L.call('keep_layernorm', x, gamma, beta, res, out, pos, 0 if pos is None else pos.shape[0], out2, M, C, float(eps))
        return out if pos is None else (out, out2)

    @staticmethod
    def geglu(x):
        F = x.shape[-1] // 2
        M = x.numel() // (2 * F)
        out = empty((*x.shape[:-1], F), x)
        L.call('keep_geglu', x, out, M, F)
        return out

    def _amax_slots(self, N, like):
        """N result words for a fused max|out|: from the per-forward arena (zeroed once in begin_forward) when it has room."""
        if self.amax_arena is not None and self.amax_pos + N <= self.amax_arena.numel():
            out = self.amax_arena[self.amax_pos:self.amax_pos + N]
            self.amax_pos += N
            return out, 1
        return torch.empty((N,), dtype=torch.float32, device=like.device), 0

    def layernorm_amax(self, x, gamma, beta, *, n_img, res=None, eps=1e-5):
        """``layernorm`` (+res) and the per-image max |out| of the result in one launch: (y, amax [n_img]) -- the x3 range scale of the
        GEMM that reads y (bit-identical y; amax == ``absmax(y)``)."""
        C = x.shape[-1]
        M = x.numel() // C
        out = torch.empty_like(x)
        am, zeroed = self._amax_slots(n_img, x)
        L.call('keep_layernorm_amax', x, gamma, beta, res, out, M, C, float(eps), M // n_img, am, zeroed)
        return out, am

    def geglu_amax(self, x, *, n_img):
        """``geglu`` and the per-image max |out|: (y, amax [n_img])."""
        F = x.shape[-1] // 2
        M = x.numel() // (2 * F)
        out = empty((*x.shape[:-1], F), x)
        am, zeroed = self._amax_slots(n_img, x)
        L.call('keep_geglu_amax', x, out, n_img, M // n_img, F, am, zeroed)
        return out, am

    # ------------------------------------------------------------------ keep_attention
    def attention(self, q, k, v, o, *, B, H, Lq, Lk, D, Dv, scale, q_str, k_str, v_str, o_str, mode=0, T=0, seg_len=0,
                  img_h=0, img_w=0, ksplit=0, shift=0, kv_rot=0, n_img=0, mma=None, probe=False, amax=None):
        """Strides are (batch, token, head) element strides.  ``probe=True`` (x3 policy, mode 0): q / k / v are projections
        of an un-normalised tensor -- their ranges are probed and the kernel rescales them into the fp16 window."""
        mma = self.attn_mma if mma is None else mma
        in_dtype = L.F32
        if q.dtype == torch.bfloat16:
            assert k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16
            in_dtype, mma = L.BF16, L.MMA_BF16
        elif mma != L.MMA_F32 and (D % 16 or any(s % 4 for s in (*q_str, *k_str))):
            mma = L.MMA_F32
        given = amax
        amax = (None, None, None)
        if given is not None and all(t is not None and t.numel() == B for t in given) and mma == L.MMA_X3 and mode == 0 and in_dtype == L.F32:
            amax = tuple(given)      # per-batch upper bounds of max |q|, |k|, |v| the caller already holds (producers' fused maxima)
        elif probe and mma == L.MMA_X3 and mode == 0 and in_dtype == L.F32:
            # rows of batch b: tokens x (H heads x D) starting at b*bs; heads are contiguous slices of one row here
            amax = (absmax(q, B, Lq, H * D, q_str[1], q_str[0], self), absmax(k, B, Lk, H * D, k_str[1], k_str[0], self),
                    absmax(v, B, Lk, H * Dv, v_str[1], v_str[0], self))
        if DEBUG_SYNC:
            import sys
            print(f'[keep] attention mma={mma} in_dtype={in_dtype} B={B} H={H} Lq={Lq} Lk={Lk} D={D} Dv={Dv} mode={mode}',
                  file=sys.stderr, flush=True)
        L.attention(q=q, k=k, v=v, o=o,
                    q_bs=q_str[0], q_ts=q_str[1], q_hs=q_str[2], k_bs=k_str[0], k_ts=k_str[1], k_hs=k_str[2],
                    v_bs=v_str[0], v_ts=v_str[1], v_hs=v_str[2], o_bs=o_str[0], o_ts=o_str[1], o_hs=o_str[2],
                    B=B, H=H, Lq=Lq, Lk=Lk, D=D, Dv=Dv, scale=float(scale), mode=mode, T=T, seg_len=seg_len,
                    img_h=img_h, img_w=img_w, ksplit=ksplit, shift=shift, kv_rot=kv_rot, n_img=n_img, mma=mma,
                    in_dtype=in_dtype, q_amax=amax[0], k_amax=amax[1], v_amax=amax[2], flags=self.attn_flags)
        if DEBUG_SYNC:
            torch.cuda.synchronize()
            if not bool(torch.isfinite(o).all()):
                import sys
                print(f'[keep] NON-FINITE attention output; q/k/v absmax {float(q.float().abs().max()):.4g} '
                      f'{float(k.float().abs().max()):.4g} {float(v.float().abs().max()):.4g}', file=sys.stderr, flush=True)
        return o

    def token_linear(self, x, w, bias=None, out_bf16=False):
        """Streaming GEMM for the GMFlow projections (bf16 policy): x [M,128] fp32 @ w[N,128]^T, N in {128,256,384}."""
        K = x.shape[-1]
        M = x.numel() // K
        N = w.shape[0]
        out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
        L.call('keep_token_linear', x, self.bf16_twin(w), bias, out, M, K, N, L.BF16 if out_bf16 else L.F32)
        return out

    def gm_mlp(self, a, b, w0, w2):
        """GMFlow FFN fused (bf16 policy): W2 . gelu(W0 . cat[a | b]); a, b [M,C] fp32, w0 [8C,2C], w2 [C,8C] fp32 views of
        the packed blob (their bf16 twins are used)."""
        C = a.shape[-1]
        M = a.numel() // C
        out = empty((M, C), a)
        L.call('keep_gm_mlp', a, b, self.bf16_twin(w0), self.bf16_twin(w2), out, M, C)
        return out


def absmax(x, N, R, C, ld, img_stride, ops=None):
    """Range probe: max |x| per image over R rows x C columns (row stride ld, image stride img_stride) -> [N] floats.
    ``ops``: take the result slots from that Ops' per-forward arena (already zero: no zero-fill launch)."""
    if ops is not None and ops.amax_arena is not None and ops.amax_pos + N <= ops.amax_arena.numel():
        out = ops.amax_arena[ops.amax_pos:ops.amax_pos + N]
        ops.amax_pos += N
        L.call('keep_absmax', x, out, N, R, C, ld, img_stride, 1)
        return out
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    L.call('keep_absmax', x, out, N, R, C, ld, img_stride, 0)
    return out


def concat2(a, b, pad_to=1):
    """cat([a, b], -1); ``pad_to``: zero-pad the channel count up to a multiple (a 130-channel conv input becomes 144 wide so
    that the 16-channel-chunk MFMA kernels take it; the weights carry matching zero columns, engine/weights.py)."""
    C1, C2 = a.shape[-1], b.shape[-1]
    M = a.numel() // C1
    ld = (C1 + C2 + pad_to - 1) // pad_to * pad_to
    out = empty((*a.shape[:-1], ld), a)
    L.call('keep_concat2', a, b, out, M, C1, C2, ld)
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


def rgb_s2d(x):
    """[N,3,H,W] in [-1,1] -> [N,H/2,W/2,16]: GMFlow's input normalisation + 2x2 space-to-depth (keep_rgb_s2d)."""
    N, C, H, W = x.shape
    assert C == 3 and H % 2 == 0 and W % 2 == 0
    out = empty((N, H // 2, W // 2, 16), x)
    L.call('keep_rgb_s2d', x, out, N, H, W)
    return out


def nhwc_to_nchw(x):
    N, H, W, C = x.shape
    out = empty((N, C, H, W), x)
    L.call('keep_nhwc_to_nchw', x, out, N, C, H * W)
    return out


def split_x3(w2d, scale):
    """[rows, Cin] fp32 (Cin % 16 == 0) -> int16 [rows, Cin/16, 2, 16]: fp16 bit patterns of hi = fp16(w*scale) and
    lo = fp16(w*scale - hi) per 16-channel chunk -- the `weight_x3` layout of keep_conv2d (csrc/keep_conv_x3.hip)."""
    rows, cin = w2d.shape
    ws = w2d.float() * scale
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    out = torch.stack((hi.view(rows, cin // 16, 16), lo.view(rows, cin // 16, 16)), dim=2)
    return out.contiguous().view(torch.int16)


FFN_W2_PERM = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def ffn_w2_perm(w2):
    """[C, hidden] -> the same matrix with every group of 16 hidden units reordered [0-3, 8-11, 4-7, 12-15]: the order in which the
    accumulator layout of v_mfma_f32_32x32x16_f16 hands a lane (half g) its hidden units {4g + (r & 3) + 8 (r >> 2)} -- the K order of
    the second product inside keep_gm_ffn_x3 (csrc/keep_ffn_x3.hip)."""
    C, Hd = w2.shape
    idx = torch.tensor(FFN_W2_PERM, device=w2.device)
    return w2.float().view(C, Hd // 16, 16).index_select(2, idx).reshape(C, Hd).contiguous()


def up2_phase_weights(w):
    """Packed [Cout,3,3,Cin] fp32 -> [4,Cout,3,3,Cin]: the 3x3 kernel of `nearest x2 -> conv3x3(pad 1)` (VQ:146-156) as four phase
    kernels on the SOURCE grid.  Output pixel (2y+py, 2x+px) reads upsampled rows 2y+py-1 .. 2y+py+1 = source rows (y-1, y, y) for
    py = 0 and (y, y, y+1) for py = 1: taps that meet the same source pixel are added (fp32), the freed taps are zero -- phase
    (py, px) keeps taps kh in {py, py+1}, kw in {px, px+1} of a 3x3 window centred on source pixel (y, x)."""
    r = torch.zeros(2, 3, 3, dtype=w.dtype, device=w.device)          # r[p][new][old]
    r[0, 0, 0] = 1; r[0, 1, 1] = 1; r[0, 1, 2] = 1                     # phase 0: new0 = old0, new1 = old1 + old2
    r[1, 1, 0] = 1; r[1, 1, 1] = 1; r[1, 2, 2] = 1                     # phase 1: new1 = old0 + old1, new2 = old2
    w4 = torch.einsum('pak,qbl,oklc->pqoabc', r, r, w.float())
    return w4.reshape(4, w.shape[0], 3, 3, w.shape[3]).contiguous()


def make_x3_blob(dev_blob, index, weights, names):
    """Split-fp16 twin of the tensors ``names`` of a packed fp32 blob (``index``: name -> (element offset, shape); ``weights``:
    name -> device view) with ONE POWER-OF-TWO SCALE PER TENSOR: each tensor's largest |w| lands just below 2^15, so a tensor of
    small weights keeps normal `lo` halves whatever the largest weight elsewhere in the net is (one shared scale -- rounds 2 / 3 --
    let a single large tensor push every other layer's `lo` halves towards the fp16 subnormals).  Returns (int16 blob with two
    elements per weight, [(first element, one past the last, accumulator scale 2^-e), ...] sorted by offset) for
    ``Ops.set_precision(..., x3_scales=)``."""
    bx = torch.zeros(2 * dev_blob.numel(), dtype=torch.int16, device=dev_blob.device)
    table = []
    for n in names:
        off, shape = index[n]
        t = weights[n]
        sc = x3_scale_for(float(t.abs().max()))
        bx[2 * off:2 * (off + t.numel())] = split_x3(t.reshape(-1, shape[-1]), sc).view(-1)
        table.append((int(off), int(off + t.numel()), 1.0 / sc))
    table.sort()
    return bx, table


def x3_scale_for(max_abs):
    """Power of two that puts the largest weight just below 2^15 (fp16 max 65504), so that small weights keep a normal lo."""
    import math
    if not max_abs > 0:
        return 1.0
    return float(2.0 ** (14 - math.ceil(math.log2(max_abs))))


# default instance: fp32 policy, no blobs -- kernel tests pass mma= / wb= / wx3= explicitly
DEFAULT = Ops()
conv = DEFAULT.conv
linear = DEFAULT.linear
norm_affine = Ops.norm_affine
layernorm = Ops.layernorm
geglu = Ops.geglu
attention = DEFAULT.attention
token_linear = DEFAULT.token_linear
gm_mlp = DEFAULT.gm_mlp
