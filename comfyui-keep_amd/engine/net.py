"""KeepNet -- the MI355X engine behind ``keep_model.keep_net``: ``KEEP.forward``
(reference keep_arch.py:1008-1145, eval branch) as a sequence of hand-written gfx950 kernels.

Contract kept from the reference ``nn.Module``: ``KeepNet(**arch)``,
``load_state_dict(sd, strict=True)``, ``.to(device)``, ``.eval()``, ``.parameters()``,
``__call__(x[B,T,3,H,W] fp32 device tensor in [-1,1], need_upscale=False) -> same shape, unclamped``.

Design: activations channels-last (feature maps [N,H,W,C] == token matrices [N*H*W,C],
so every ``rearrange`` of the reference is free); GroupNorm/InstanceNorm are applied lazily as a
per-(image,channel) affine in the consuming convolution's prologue; attention never materialises
scores; window partition / roll / key concatenation / [f0;f1] swap are index math inside the
attention kernel.  B independent clips ride the batch axis of every kernel.
"""
import math
import os

import numpy as np
import torch

from . import hiplib as L
from . import ops
from .arch import (CHANNELS, DEFAULT_ARCH, FUSE_ENCODER_BLOCK, FUSE_GENERATOR_BLOCK, GMFLOW, encoder_blocks,
                   generator_blocks)
from .weights import logical_tensors, pack_blob, validate_state_dict, views

_KNOWN_KW = set(DEFAULT_ARCH) | {
    'gumbel_straight_through', 'gumbel_kl_weight', 'vqgan_path', 'fix_modules', 'flownet_path', 'cfa_nlayers',
    'cross_residual', 'mask_ratio'}


FUSED_GM_MLP = os.environ.get('KEEP_NO_FUSED_MLP') is None     # dev switch
FUSED_GM_FFN_X3 = os.environ.get('KEEP_X3_FUSED_FFN', '1') != '0'   # x3 policy: GMFlow mlp.0 + GELU + mlp.2 + norm2 + residual as one kernel (A/B: 0)
GM_S2D_CONV1 = os.environ.get('KEEP_GM_S2D', '1') != '0'         # GMFlow conv1 (7x7 s2) as a 4x4 convolution on the space-to-depth image (A/B: 0)
GM_DEDUP_L0 = os.environ.get('KEEP_GM_DEDUP_L0', '1') != '0'    # GMFlow layer-0 self-attention once per frame instead of once per pair member (A/B: 0)
# 'x3': split-fp16 operands on the 16-bit matrix pipe (fp32-grade products, csrc/keep_conv_x3.hip) -- the default: it
# passes the same <= 1e-3 parity tests as 'fp32' (exact f32 MFMA everywhere) at several times its speed.
CHECK_X3_RANGE = os.environ.get('KEEP_X3_NO_RANGE_CHECK') is None
GRAPH_MAX_CLIPS = int(os.environ.get('KEEP_AMD_GRAPH_MAX_CLIPS', '2'))
# two-stream order of the forward (GMFlow + Kalman gains on a second stream under the frame recurrence) for calls of at most this
# many clips (0 = never); the first chunk of pairs / the following chunks (pairs per GMFlow launch group)
STREAM_OVERLAP_MAX_CLIPS = int(os.environ.get('KEEP_AMD_OVERLAP_MAX_CLIPS', '2'))
# CFT: the encoder half of encode_enc's first convolution and shortcut once per clip for all frames, the decoder half in the frame loop (A/B: 0
# = the reference's order: one convolution over cat[enc, dec] per frame)
CFT_SPLIT = os.environ.get('KEEP_CFT_SPLIT', '1') != '0'
CFA_FREE_RANGES = os.environ.get('KEEP_CFA_FREE_RANGES', '1') != '0'   # x3: CFA range scales from producers' fused maxima instead of probes (A/B: 0)
# the CFA block's LayerNorm / GEGLU maxima fused into their kernels up to this many token rows per launch (few clips in flight: the probe launch is
# the cost); above it the probes stay -- thousands of blocks behind one atomic word per image cost more than the probe (+4 ms per 16-clip step,
# tools/dev/cfa_ab.py).  The maxima are exact either way: same bits.
CFA_FUSED_AMAX_ROWS = int(os.environ.get('KEEP_CFA_FUSED_AMAX_ROWS', '4096'))
STREAM_OVERLAP_FIRST = int(os.environ.get('KEEP_AMD_OVERLAP_FIRST', '3'))
STREAM_OVERLAP_CHUNK = int(os.environ.get('KEEP_AMD_OVERLAP_CHUNK', '4'))
GRAPH_CACHE = 4
RESIDENT = os.environ.get('KEEP_AMD_RESIDENT', '0') == '1'      # keep the packed weights on the device across offload()
PRECISIONS = ('fp32', 'x3', 'bf16')
DEFAULT_PRECISION = 'x3'


ROCTX = os.environ.get('KEEP_AMD_ROCTX', '0') == '1'      # per-stage roctx ranges (rocprofv3 --marker-trace / --kernel-trace timelines)


class _Range:
    """``with _Range('K2 lq_encoder'):`` -- a roctx range around one stage of the forward (torch.cuda.nvtx is roctx on ROCm);
    off by default: a range is a host-side marker, but ~90 of them per clip are not free at B = 1."""
    __slots__ = ('name',)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if ROCTX:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if ROCTX:
            torch.cuda.nvtx.range_pop()
        return False


class _FrameList:
    """A clip handed to run_clips_u8 as T separate uint8 [H,W,3] crops (what the processor holds): behaves like a [T,H,W,3]
    tensor for shape checks and is copied crop by crop into the pinned upload buffer -- no intermediate np.stack."""
    __slots__ = ('frames', 'shape', 'dtype')

    def __init__(self, frames):
        self.frames = [np.ascontiguousarray(f) for f in frames]
        f0 = self.frames[0]
        if any(f.shape != f0.shape or f.dtype != np.uint8 for f in self.frames) or f0.ndim != 3:
            raise ValueError("a clip given as a list must hold uint8 [H,W,3] crops of one size")
        self.shape = (len(self.frames),) + tuple(f0.shape)
        self.dtype = torch.uint8


class KeepNet:
    def __init__(self, **arch):
        unknown = set(arch) - _KNOWN_KW
        if unknown:
            raise TypeError(f"KeepNet got unexpected architecture keys {sorted(unknown)}")
        self.cfg = dict(DEFAULT_ARCH, **{k: v for k, v in arch.items() if k in DEFAULT_ARCH})
        if self.cfg['quantizer_type'] != 'nearest':
            raise NotImplementedError("only the 'nearest' quantizer is on the KEEP inference path")
        if arch.get('cross_residual', True) is not True:
            raise NotImplementedError("cross_residual=False is not used by any shipped KEEP config")
        self._sd = None            # reference-layout fp32 CPU tensors (what load_state_dict received)
        self._blob = None          # packed numpy blob + index
        self._index = None
        self._dev_blob = None      # device copy
        self.w = None              # name -> device view
        self.device = torch.device('cpu')
        self.training = False
        self._const = {}           # per-shape device constants (position tables, grids)
        self.last_aux = None
        self._dev_blob16 = None    # bf16 twin of the packed blob (same offsets) for the bf16-MFMA policy
        self._dev_blobx3 = None    # split-fp16 twin (2 x int16 per weight) for the x3 policy
        self._x3_scales = None     # per-tensor accumulator scales of the x3 twin (ops.make_x3_blob)
        self.o = ops.Ops()         # this net's precision policy + weight twins (never shared between nets)
        self.x3_fallbacks = 0      # batches the x3 policy handed back to the f32 kernels (non-finite output)
        # hipGraph replay of the whole forward for small batches (launch-bound: ~9 k kernels per clip): 'auto' = at most
        # GRAPH_MAX_CLIPS clips per call, '1' = always, '0' = never.  One captured graph per (B, T, H, W, policy).
        self.graph_mode = os.environ.get('KEEP_AMD_GRAPH', 'auto')
        self._graphs = {}
        self._graph_seen = {}      # graph key -> eager occurrences so far
        # With an initialised process group, ONE clip list handed to run_clips_u8 on every rank is sharded over the ranks
        # (True, default).  False: every rank restores the list IT is given (one video per GPU, BASELINE configs[4]).
        self.shard_across_ranks = os.environ.get('KEEP_AMD_SHARD', '1') == '1'
        self.pool = None           # engine/pool.py:GpuPool when this process drives several GPUs (KEEP_AMD_GPUS=N)
        self._aux_top1 = []
        self._pinned_in = {}       # pinned upload staging buffers of run_clips_u8, by (shape, slot)
        self._pinned = None        # pinned host copy of the packed blob (made at the first upload)
        self.weights_generation = 0   # bumped whenever the packed blob changes (load_state_dict / adopt_packed): a worker pool that
                                      # received an older blob is stale and is closed (engine/pool.py)
        self.precision = 'fp32'
        self.set_precision(os.environ.get('KEEP_AMD_PRECISION', DEFAULT_PRECISION))

    # ------------------------------------------------------------------ nn.Module-like surface
    def load_state_dict(self, state_dict, strict=True):
        if strict:
            validate_state_dict(state_dict, self.cfg)
        self._sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}
        self._blob, self._index = pack_blob(logical_tensors(self._sd, self.cfg))
        self._dev_blob, self.w, self._pinned = None, None, None
        self.weights_generation += 1
        self.close_pool()           # its workers hold the previous weights; .to('cuda') starts a fresh one (KEEP_AMD_GPUS)
        if self.device.type == 'cuda':
            self._upload()
            self._maybe_start_pool()
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def parameters(self):
        return iter((self._sd or {}).values())

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("KeepNet is an inference engine (the reference trains with BasicSR)")
        return self.eval()

    supports_sink = True      # run_clips_u8(sink=...): finished groups are handed over on the GPU (the processor's streamed paste-back)

    # frame 0 of a clip depends on no other frame (no flow, no Kalman update, no CFA for i == 0), so a lone crop can be
    # restored as a T = 1 clip instead of the reference's T = 2 duplicate (keep_processor.py:173-178) with the same result
    supports_single_frame = True

    def set_precision(self, precision):
        """'fp32': exact f32 MFMA everywhere.  'x3': every matrix-core operand split into two fp16 halves, three MFMAs per
        product (<= 2^-22 relative per product, fp32 accumulate) -- parity-grade (<= 1e-3) at the 16-bit pipe's rate / 3.
        'bf16': operands rounded to bf16 (speed policy, outside the parity tolerance)."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
        self.precision = precision
        return self

    def _make_x3(self):
        """Split-fp16 twin of every matrix weight in the blob (2-D+ tensors whose reduction axis is a multiple of 16)."""
        # only tensors a keep_conv2d call consumes as its x3 operand: the position table and the codebook are read by
        # elementwise / gather kernels (a large-magnitude table must not shrink the scale of every convolution weight)
        names = [n for n, (_, shape) in self._index.items() if len(shape) >= 2 and shape[-1] % 16 == 0
                 and n not in ('position_emb', 'quantize.embedding.weight')]
        # one power-of-two scale PER TENSOR (the ABI carries x3_acc_scale per launch): a tensor with one 100x weight does not
        # cost every other layer its `lo` bits (tests/test_gpu_net.py::test_x3_scale_is_per_tensor)
        self._dev_blobx3, self._x3_scales = ops.make_x3_blob(self._dev_blob, self._index, self.w, names)

    def _activate_precision(self):
        if self.precision == 'bf16':
            if self._dev_blob16 is None:
                self._dev_blob16 = self._dev_blob.to(torch.bfloat16)
            self.o.set_precision(L.MMA_BF16, self._dev_blob, self._dev_blob16)
        elif self.precision == 'x3':
            if self._dev_blobx3 is None:
                self._make_x3()
            self.o.set_precision(L.MMA_X3, self._dev_blob, None, self._dev_blobx3, 1.0, x3_scales=self._x3_scales)
            if FUSED_GM_FFN_X3:     # permuted x3 twins of the GMFlow mlp.2 weights (keep_gm_ffn_x3): built here, never inside a stream capture
                for n_ in self.w:
                    if n_.startswith('flownet.') and n_.endswith('.mlp.2.weight') and self.w[n_].shape[0] == 128:
                        self.o.ffn_w2_twin(self.w[n_])
            if ops.UP2_PHASES:      # phase kernels of the generator's Upsample convolutions: built here, never inside a stream capture
                for i, (kind, _, _) in enumerate(generator_blocks(self.cfg)):
                    if kind == 'up':
                        self.o.up2_twin(self.w[f'generator.blocks.{i}.conv.weight'])
        else:
            self.o.set_precision(L.MMA_F32, self._dev_blob, None)

    def _upload(self, from_blob=None):
        L.load(check_device=True)       # fails loudly: no library / not gfx950 -> no silent fallback
        if from_blob is None:
            if self._pinned is None:                    # one pinned staging copy: re-uploads run at PCIe rate
                self._pinned = torch.from_numpy(self._blob).pin_memory()
            from_blob = self._pinned
        self._dev_blob = from_blob.to(self.device, non_blocking=False)
        self._dev_blob16 = self._dev_blobx3 = None
        self._graphs = {}
        self.w = views(self._dev_blob, self._index)

    def to(self, device):
        device = torch.device(device)
        if device.type == 'cuda':
            if device.index is None:
                device = torch.device('cuda', torch.cuda.current_device())
            changed = (self.device != device) or self._dev_blob is None
            self.device = device
            if self._blob is not None and changed:
                with torch.cuda.device(device):
                    self._upload()
            self._maybe_start_pool()
        elif RESIDENT and self._dev_blob is not None:
            # residency policy: KEEPModelPack.offload() after every node call would drop 633 MB of packed
            # weights (plus the policy's twin) and the next call would upload and re-derive them; with KEEP_AMD_RESIDENT=1
            # the device copy is parked instead -- .to('cuda') finds it in place (288 GB of HBM: nothing else wants it)
            pass
        else:
            self.device = device
            # the worker pool (KEEP_AMD_GPUS) stays up: its workers hold THEIR copies of the same weights (weights_generation guards a
            # stale one; load_state_dict / adopt_packed close it) -- KEEPModelPack.offload() runs after every node call, and closing the
            # pool here cost every call a respawn of the workers (torch import, HIP init), a 633 MB broadcast and the x3 twins
            self._dev_blob, self._dev_blob16, self._dev_blobx3, self.w = None, None, None, None
            self.o.set_precision(self.o.mma)        # drop this net's references to the device blobs
            self._const = {}
            self._graphs = {}
        return self

    def _maybe_start_pool(self):
        """KEEP_AMD_GPUS=N: N - 1 worker processes, one weight broadcast -- whenever weights are resident and no pool is up."""
        if self.pool is None and self._dev_blob is not None:
            from . import pool as kpool
            if kpool.wanted_gpus() > 1 and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
                self.start_pool(kpool.wanted_gpus())

    def start_pool(self, n_gpus):
        """Drive ``n_gpus`` GPUs of this node from THIS process (engine/pool.py): spawns n_gpus - 1 workers, broadcasts the packed
        weights to them once (RCCL over xGMI), and makes ``run_clips_u8`` shard its clips over all of them."""
        from .pool import GpuPool
        if self.pool is not None:
            self.pool.close()
        self.pool = GpuPool(self, n_gpus)
        return self.pool

    def close_pool(self):
        """Stop the worker processes of this net's pool (they hold a copy of the weights in their GPUs' HBM)."""
        pool, self.pool = self.pool, None
        if pool is not None:
            pool.close()

    def pool_config(self):
        """Everything a pool worker must share with the root for `bit for bit equal to the sequential loop` to hold: the precision
        policy, the plans' reference batch, the kernel-selection overrides and the hipGraph mode.  ``GpuPool.run`` compares it with
        what the workers were last told and re-configures them when it moved (``set_precision`` after the pool came up)."""
        return {'precision': self.precision, 'plan_ref_images': int(self.o.plan_ref_images), 'flags': int(self.o.flags),
                'attn_flags': int(self.o.attn_flags), 'graph_mode': str(self.graph_mode)}

    def apply_pool_config(self, cfg):
        """Worker side of ``pool_config``."""
        self.set_precision(cfg['precision'])
        self.o.plan_ref_images, self.o.flags, self.o.attn_flags = int(cfg['plan_ref_images']), int(cfg['flags']), int(cfg['attn_flags'])
        self.graph_mode = str(cfg['graph_mode'])

    def packed_blob(self):
        """Device blob (for the RCCL weight broadcast, engine/dist.py)."""
        return self._dev_blob

    def adopt_packed(self, index, dev_blob):
        """Install a packed blob received from another rank."""
        self._index, self._dev_blob, self._dev_blob16, self._dev_blobx3 = index, dev_blob, None, None
        self._graphs = {}
        self.weights_generation += 1
        self.close_pool()
        self.device = dev_blob.device
        self.w = views(dev_blob, index)

    # ------------------------------------------------------------------ building blocks (VQGAN)
    def _gn(self, x, p, st=None):
        """GroupNorm(32, eps 1e-6) of x as a conv prologue; ``st`` = statistics from the producing conv's epilogue."""
        return ops.norm_affine(x, self.w[f'{p}.weight'], self.w[f'{p}.bias'], 32, 1e-6, stats=st)

    def _resblock(self, x, p, st=None):
        """VQ:170-181.  (x, statistics of x or None) -> (y, statistics of y)."""
        w = self.w
        # bf16 policy: h is read once, by the normalise+swish pass in front of conv2's halo kernel -> store it as bf16
        # (its GroupNorm statistics come from conv1's epilogue, taken on the fp32 values); per-image rule (maps of at
        # least 128x128), so results do not depend on the batch size.  The library refuses where it cannot (plan).
        h16 = self.o.mma == L.MMA_BF16 and x.shape[1] * x.shape[2] >= 16384
        h, hst = self.o.conv(x, w[f'{p}.conv1.weight'], w[f'{p}.conv1.bias'], pro=self._gn(x, f'{p}.norm1', st),
                             pro_act=L.PRO_SWISH, stats=True, out_bf16=h16)
        sc = x
        if f'{p}.conv_out.weight' in w:
            # (round 5 ran this 1x1 shortcut on a second stream next to conv1 -- it reads only x: 1.3 % SLOWER at 16 clips, nothing at
            # one; the same for the q|k and v projections of the code transformer and q / kv of CFA: profiles/r05_small_forks_ab.txt)
            sc = self.o.linear(x, w[f'{p}.conv_out.weight'], w[f'{p}.conv_out.bias'], n_img=x.shape[0],
                               x_amax=None if st is None else st.amax)
        return self.o.conv(h, w[f'{p}.conv2.weight'], w[f'{p}.conv2.bias'], pro=self._gn(h, f'{p}.norm2', hst),
                           pro_act=L.PRO_SWISH, residual=sc, stats=True)

    def _attnblock(self, x, p, st=None):
        """VQ:219-243: GN -> q,k,v (one GEMM) -> fused attention (1 head, d=C) -> proj_out + x."""
        w = self.w
        N, H, Wd, C = x.shape
        HW = H * Wd
        qkv = self.o.linear(x.view(N * HW, C), w[f'{p}.qkv.weight'], w[f'{p}.qkv.bias'], pro=self._gn(x, f'{p}.norm', st),
                            n_img=N, out_bf16=True)
        o = ops.empty((N * HW, C), x)
        s3 = (HW * 3 * C, 3 * C, 0)
        self.o.attention(qkv, ops.offset(qkv, C), ops.offset(qkv, 2 * C), o, B=N, H=1, Lq=HW, Lk=HW, D=C, Dv=C,
                      scale=int(C) ** (-0.5), q_str=s3, k_str=s3, v_str=s3, o_str=(HW * C, C, 0))
        y = self.o.linear(o, w[f'{p}.proj_out.weight'], w[f'{p}.proj_out.bias'], residual=x.view(N * HW, C), bounded=True, n_img=N)
        return y.view(N, H, Wd, C)

    def _vq_stack(self, x, prefix, blocks, taps=(), hook=None, tap_stats=None):
        """Encoder.forward / Generator.forward (VQ:288-292, 339-343) over NHWC maps.
        ``hook(j, x, st) -> (x, st)`` runs after block j (generator CFT/CFA taps, KA:1104-1121).  ``st`` travels with x:
        the per-channel (sum, sumsq) partials its producing convolution reduced in the epilogue, or None."""
        w = self.w
        feats = {}
        st = None
        pending = None            # a bare GroupNorm block folds into the next conv's prologue (no swish)
        for i, (kind, _, _) in enumerate(blocks):
            p = f'{prefix}.blocks.{i}'
            if kind == 'conv':
                x, st = self.o.conv(x, w[f'{p}.weight'], w[f'{p}.bias'], pro=pending, stats=True)
                pending = None
            elif kind == 'res':
                x, st = self._resblock(x, p, st)
            elif kind == 'attn':
                x, st = self._attnblock(x, p, st), None
            elif kind == 'down':
                x, st = self.o.conv(x, w[f'{p}.conv.weight'], w[f'{p}.conv.bias'], down=True, stats=True,
                                    x_amax=None if st is None else st.amax)
            elif kind == 'up':
                x, st = self.o.conv(x, w[f'{p}.conv.weight'], w[f'{p}.conv.bias'], upsample=True, stats=True,
                                    x_amax=None if st is None else st.amax)
            elif kind == 'norm':
                pending = self._gn(x, p, st)
            if i in taps:
                feats[str(x.shape[2])] = x
                if tap_stats is not None:                # (what the producing convolution reduced about the tap: GroupNorm partials, max|x|)
                    tap_stats[str(x.shape[2])] = st
            if hook is not None and kind != 'norm':
                x, st = hook(i, x, st)
        return x, feats

    # ------------------------------------------------------------------ code prediction (KA:1073-1089)
    def _predict_codes(self, z_hat, force_idx=None, want_aux=False):
        w, cfg = self.w, self.cfg
        B = z_hat.shape[0]
        Ltok = z_hat.shape[1] * z_hat.shape[2]
        D, nh = cfg['dim_embd'], cfg['n_head']
        dh = D // nh
        q = self.o.linear(z_hat.view(B * Ltok, -1), w['feat_emb.weight'], w['feat_emb.bias'], n_img=B)
        pos = w['position_emb']
        for i in range(cfg['n_layers']):
            p = f'ft_layers.{i}'
            x2, qk_in = ops.layernorm(q, w[f'{p}.norm1.weight'], w[f'{p}.norm1.bias'], pos=pos)
            wi, bi = w[f'{p}.self_attn.in_proj_weight'], w[f'{p}.self_attn.in_proj_bias']
            qk = self.o.linear(qk_in, wi[:2 * D], bi[:2 * D], out_bf16=True, bounded=True, n_img=B)
            v = self.o.linear(x2, wi[2 * D:], bi[2 * D:], out_bf16=True, bounded=True, n_img=B)
            o = ops.empty((B * Ltok, D), q)
            self.o.attention(qk, ops.offset(qk, D), v, o, B=B, H=nh, Lq=Ltok, Lk=Ltok, D=dh, Dv=dh, scale=dh ** -0.5,
                          q_str=(Ltok * 2 * D, 2 * D, dh), k_str=(Ltok * 2 * D, 2 * D, dh),
                          v_str=(Ltok * D, D, dh), o_str=(Ltok * D, D, dh))
            q = self.o.linear(o, w[f'{p}.self_attn.out_proj.weight'], w[f'{p}.self_attn.out_proj.bias'], residual=q, bounded=True, n_img=B)
            x2 = ops.layernorm(q, w[f'{p}.norm2.weight'], w[f'{p}.norm2.bias'])
            h = self.o.linear(x2, w[f'{p}.linear1.weight'], w[f'{p}.linear1.bias'], act=L.ACT_GELU, bounded=True, n_img=B)
            q = self.o.linear(h, w[f'{p}.linear2.weight'], w[f'{p}.linear2.bias'], residual=q, bounded=True, n_img=B)
        xl = ops.layernorm(q, w['idx_pred_layer.0.weight'], w['idx_pred_layer.0.bias'])
        logits = self.o.linear(xl, w['idx_pred_layer.1.weight'], bounded=True, n_img=B)
        cb = w['quantize.embedding.weight']
        quant = ops.empty((B * Ltok, cb.shape[1]), q)
        idx = torch.empty((B * Ltok,), dtype=torch.int32, device=q.device)
        margin = ops.empty((B * Ltok,), q) if want_aux else None
        # a non-finite logit row (an fp16-range overflow anywhere on hq_encoder -> Kalman update -> transformer) must not
        # become a plausible code: the kernel raises the forward's status word and NaN-fills the row (keep_hip.h)
        L.call('keep_argmax_gather', logits, cb, force_idx, idx, margin, quant, B * Ltok, cb.shape[0], cb.shape[1], self.o.status)
        side = z_hat.shape[1]
        if want_aux:
            self._aux_top1.append(logits.view(B, Ltok, -1).max(-1).values)
        return quant.view(B, side, z_hat.shape[2], cb.shape[1]), idx.view(B, Ltok), \
            (None if margin is None else margin.view(B, Ltok))

    # ------------------------------------------------------------------ CFT / CFA
    def _cft_splittable(self, p, C):
        w = self.w
        return (CFT_SPLIT and f'{p}.encode_enc.conv1.weight_enc' in w and w[f'{p}.encode_enc.conv1.weight_enc'].shape[-1] == C
                and self.o.mma != L.MMA_BF16)

    def _cft_enc_part(self, enc, p, st=None):
        """The encoder half of ``encode_enc`` (VQ:170-181 over cat[enc, dec], KA:466): GroupNorm(32 groups over 2C channels) never mixes the
        two halves -- groups 0-15 are GroupNorm(16 groups) of enc with norm1's first C weights -- and conv1 / the 1x1 shortcut are sums over
        input channels, so   conv1(swish(GN(cat)))  = conv1[:, :C](swish(GN16(enc))) + conv1[:, C:](swish(GN16(dec))) + bias   and
        conv_out(cat) = conv_out[:, :C](enc) + conv_out[:, C:](dec) + bias.   enc [N,h,w,C] (all frames of the clips at once) ->
        (A1, S1): the two encoder-half sums without bias, added as residuals by ``_cft`` in the frame loop."""
        w, q = self.w, f'{p}.encode_enc'
        C = enc.shape[-1]
        g1 = ops.norm_affine(enc, w[f'{q}.norm1.weight'][:C], w[f'{q}.norm1.bias'][:C], 16, 1e-6, stats=st)
        a1 = self.o.conv(enc, w[f'{q}.conv1.weight_enc'], None, pro=g1, pro_act=L.PRO_SWISH)
        s1 = self.o.linear(enc, w[f'{q}.conv_out.weight_enc'], None, n_img=enc.shape[0], x_amax=None if st is None else st.amax)
        return a1, s1

    def _cft(self, enc, dec, p, pre=None, dec_st=None):
        """KA:465-472: dec + cond*(dec*scale(e) + shift(e)), e = ResBlock(cat[enc, dec]).  ``pre``: ``_cft_enc_part(enc)`` of this frame
        (computed per clip by the forward); ``dec_st``: the statistics the producer of ``dec`` reduced."""
        w = self.w
        C = dec.shape[-1]
        if self._cft_splittable(p, C):
            q = f'{p}.encode_enc'
            a1, s1 = pre if pre is not None else self._cft_enc_part(enc, p)
            g1 = ops.norm_affine(dec, w[f'{q}.norm1.weight'][C:], w[f'{q}.norm1.bias'][C:], 16, 1e-6, stats=dec_st)
            h, hst = self.o.conv(dec, w[f'{q}.conv1.weight_dec'], w[f'{q}.conv1.bias'], pro=g1, pro_act=L.PRO_SWISH, residual=a1, stats=True)
            sc = self.o.linear(dec, w[f'{q}.conv_out.weight_dec'], w[f'{q}.conv_out.bias'], residual=s1, n_img=dec.shape[0],
                               x_amax=None if dec_st is None else dec_st.amax)
            e, est = self.o.conv(h, w[f'{q}.conv2.weight'], w[f'{q}.conv2.bias'], pro=self._gn(h, f'{q}.norm2', hst), pro_act=L.PRO_SWISH,
                                 residual=sc, stats=True)
        else:
            e, est = self._resblock(ops.concat2(enc, dec), f'{p}.encode_enc')
        # (only the fused max|out| of ss is used -- its GroupNorm partials never were: 'amax' lets the 16 x 16 stage take the 64-pixel blocks)
        ss, sst = self.o.conv(e, w[f'{p}.ss0.weight'], w[f'{p}.ss0.bias'], act=L.ACT_LRELU02, stats='amax',
                              x_amax=None if est is None else est.amax)                          # [.., 2C]
        ss_amax = None if sst is None else sst.amax             # max over all 2C channels bounds either half
        scale = self.o.conv(ss, w[f'{p}.scale.2.weight'], w[f'{p}.scale.2.bias'], cin=C, in_off=0, x_amax=ss_amax)
        return self.o.conv(ss, w[f'{p}.shift.2.weight'], w[f'{p}.shift.2.bias'], cin=C, in_off=C, residual=dec,
                           aux=scale, aux_w=self.cfg['cond'], stats=True, x_amax=ss_amax)

    def _cfa(self, curr, prev, p, curr_amax=None, prev_amax=None, want_amax=False):
        """KA:519-541 (post-norm): a = attn(curr, prev); y = LN(a)+curr; LN(ff(y))+y.
        x3 range scales: none of the nine is probed when the caller supplies ``curr_amax`` / ``prev_amax`` -- ``curr_amax`` (the producing
        convolution's fused max|out|), the q / kv projections' own fused maxima (max over k AND v bounds either), |attention output| <=
        max|v|, and the fused maxima of the two LayerNorms and the GEGLU (``keep_layernorm_amax`` / ``keep_geglu_amax``: the feed-forward
        GEMMs' inputs, and -- ``want_amax``: (z, max|z| per image) -- the NEXT frame's ``prev_amax``)."""
        w, cfg = self.w, self.cfg
        B, H, Wd, C = curr.shape
        Ltok = H * Wd
        nh, dh = cfg['cfa_nhead'], cfg['cfa_dim']
        inner = nh * dh
        c = curr.view(B * Ltok, C)
        free = CFA_FREE_RANGES and self.o.mma == L.MMA_X3
        q, q_amax = self.o.linear(c, w[f'{p}.attn.to_q.weight'], out_bf16=True, n_img=B, x_amax=curr_amax if free else None, want_amax=True)
        kv, kv_amax = self.o.linear(prev.view(B * Ltok, C), w[f'{p}.attn.to_kv.weight'], out_bf16=True, n_img=B,
                                    x_amax=prev_amax if free else None, want_amax=True)
        have = free and q_amax is not None and kv_amax is not None
        o = ops.empty((B * Ltok, inner), curr)
        self.o.attention(q, kv, ops.offset(kv, inner), o, B=B, H=nh, Lq=Ltok, Lk=Ltok, D=dh, Dv=dh, scale=dh ** -0.5,
                      q_str=(Ltok * inner, inner, dh), k_str=(Ltok * 2 * inner, 2 * inner, dh),
                      v_str=(Ltok * 2 * inner, 2 * inner, dh), o_str=(Ltok * inner, inner, dh), probe=True,
                      amax=(q_amax, kv_amax, kv_amax) if have else None)
        a = self.o.linear(o, w[f'{p}.attn.to_out.0.weight'], w[f'{p}.attn.to_out.0.bias'], n_img=B, x_amax=kv_amax if have else None)
        if free and B * Ltok <= CFA_FUSED_AMAX_ROWS:
            y, y_amax = self.o.layernorm_amax(a, w[f'{p}.norm1.weight'], w[f'{p}.norm1.bias'], res=c, n_img=B)
            f, f_amax = self.o.geglu_amax(self.o.linear(y, w[f'{p}.ff.net.0.proj.weight'], w[f'{p}.ff.net.0.proj.bias'], n_img=B, x_amax=y_amax),
                                          n_img=B)
            f = self.o.linear(f, w[f'{p}.ff.net.2.weight'], w[f'{p}.ff.net.2.bias'], n_img=B, x_amax=f_amax)
            z, z_amax = self.o.layernorm_amax(f, w[f'{p}.norm2.weight'], w[f'{p}.norm2.bias'], res=y, n_img=B)
            return (z.view(B, H, Wd, C), z_amax) if want_amax else z.view(B, H, Wd, C)
        y = ops.layernorm(a, w[f'{p}.norm1.weight'], w[f'{p}.norm1.bias'], res=c)
        f = ops.geglu(self.o.linear(y, w[f'{p}.ff.net.0.proj.weight'], w[f'{p}.ff.net.0.proj.bias'], n_img=B))
        f = self.o.linear(f, w[f'{p}.ff.net.2.weight'], w[f'{p}.ff.net.2.bias'], n_img=B)
        z = ops.layernorm(f, w[f'{p}.norm2.weight'], w[f'{p}.norm2.bias'], res=y)
        return (z.view(B, H, Wd, C), None) if want_amax else z.view(B, H, Wd, C)

    # ------------------------------------------------------------------ Kalman gain (KA:801-821)
    def _kalman_gain(self, z, B, T):
        """z [B*T,h,w,C] (clip-major) -> gains [B*T, h*w]."""
        w, cfg = self.w, self.cfg
        BT, Hh, Ww, C = z.shape
        Ltok = Hh * Ww
        nh, dh = cfg['n_head'], cfg['kalman_attn_head_dim']
        inner = nh * dh
        h = z.reshape(BT * Ltok, C)
        for i in range(cfg['num_uncertainty_layers']):
            p = f'kalman_filter.uncertainty_estimator.{i}'
            # sparse-causal spatial attention (KA:686-748): keys = [frame 0 ; frame f-1]
            x1 = ops.layernorm(h, w[f'{p}.norm1.weight'], w[f'{p}.norm1.bias'])
            qkv = self.o.linear(x1, w[f'{p}.attn1.to_qkv.weight'], out_bf16=True, bounded=True, n_img=BT)
            o = ops.empty((BT * Ltok, inner), h)
            s3 = (Ltok * 3 * inner, 3 * inner, dh)
            self.o.attention(qkv, ops.offset(qkv, inner), ops.offset(qkv, 2 * inner), o, B=BT, H=nh, Lq=Ltok,
                          Lk=2 * Ltok, D=dh, Dv=dh, scale=dh ** -0.5, q_str=s3, k_str=s3, v_str=s3,
                          o_str=(Ltok * inner, inner, dh), mode=1, T=T, seg_len=Ltok)
            h = self.o.linear(o, w[f'{p}.attn1.to_out.0.weight'], w[f'{p}.attn1.to_out.0.bias'], residual=h, bounded=True, n_img=BT)
            # GEGLU feed-forward (KA:669)
            x3 = ops.layernorm(h, w[f'{p}.norm3.weight'], w[f'{p}.norm3.bias'])
            f = ops.geglu(self.o.linear(x3, w[f'{p}.ff.net.0.proj.weight'], w[f'{p}.ff.net.0.proj.bias'], bounded=True, n_img=BT))
            h = self.o.linear(f, w[f'{p}.ff.net.2.weight'], w[f'{p}.ff.net.2.bias'], residual=h, bounded=True, n_img=BT)
            # temporal attention over the T frames of each spatial token (KA:671-680): strided, no rearrange
            xt = ops.layernorm(h, w[f'{p}.norm_temp.weight'], w[f'{p}.norm_temp.bias'])
            qkv = self.o.linear(xt, w[f'{p}.attn_temp.to_qkv.weight'], out_bf16=True, bounded=True, n_img=BT)
            o = ops.empty((BT * Ltok, inner), h)
            for b in range(B):
                qb = ops.offset(qkv, b * T * Ltok * 3 * inner)
                st = (3 * inner, Ltok * 3 * inner, dh)          # batch = spatial token, token = frame
                self.o.attention(qb, ops.offset(qb, inner), ops.offset(qb, 2 * inner),
                              ops.offset(o, b * T * Ltok * inner), B=Ltok, H=nh, Lq=T, Lk=T, D=dh, Dv=dh,
                              scale=dh ** -0.5, q_str=st, k_str=st, v_str=st, o_str=(inner, Ltok * inner, dh))
            h = self.o.linear(o, w[f'{p}.attn_temp.to_out.0.weight'], w[f'{p}.attn_temp.to_out.0.bias'], residual=h, bounded=True, n_img=BT)
        m = h.view(BT, Hh, Ww, C)
        mst = None
        for i in range(3):
            m, mst = self._resblock(m, f'kalman_filter.kalman_gain_calculator.{i}', mst)
        g = self.o.linear(m.view(BT * Ltok, C), w['kalman_filter.kalman_gain_calculator.3.weight'],
                       w['kalman_filter.kalman_gain_calculator.3.bias'], act=L.ACT_SIGMOID, n_img=BT)
        return g.view(BT, Ltok)

    # ------------------------------------------------------------------ GMFlow (GF:40-66, GM/gmflow.py:92-170)
    def _gm_consts(self, h8, w8, dev):
        key = ('gm', h8, w8, str(dev))
        if key not in self._const:
            C = GMFLOW['feature_channels']
            wh, ww = h8 // 2, w8 // 2
            # GM/position.py:26-46 on a (wh x ww) window, then tiled 2x2 (GM/utils.py:66-86)
            npf = C // 2
            ye = torch.arange(1, wh + 1, dtype=torch.float32).view(wh, 1).expand(wh, ww)
            xe = torch.arange(1, ww + 1, dtype=torch.float32).view(1, ww).expand(wh, ww)
            eps, sc = 1e-6, 2 * math.pi
            ye = ye / (float(wh) + eps) * sc
            xe = xe / (float(ww) + eps) * sc
            dim_t = torch.arange(npf, dtype=torch.float32)
            dim_t = 10000.0 ** (2 * (dim_t // 2) / npf)
            px = xe[:, :, None] / dim_t
            py = ye[:, :, None] / dim_t
            px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
            py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
            win = torch.cat((py, px), dim=2)                       # [wh, ww, C]
            table = win.repeat(2, 2, 1).contiguous()               # [h8, w8, C]
            gy, gx = torch.meshgrid(torch.arange(h8), torch.arange(w8), indexing='ij')
            grid = torch.stack([gx, gy], dim=-1).float().reshape(h8 * w8, 2).contiguous()   # (x, y) per token
            self._const[key] = (table.to(dev), grid.to(dev))
        return self._const[key]

    def _inorm(self, x, st=None):
        return ops.norm_affine(x, None, None, x.shape[-1], 1e-5, stats=st)

    def _gm_resblock(self, x, p, stride):
        """GM/backbone.py:25-36."""
        w = self.w
        c1, st1 = self.o.conv(x, w[f'{p}.conv1.weight'], None, stride=stride, pad=1, stats=True, bounded=True)
        c2, st2 = self.o.conv(c1, w[f'{p}.conv2.weight'], None, pro=self._inorm(c1, st1), pro_act=L.PRO_RELU, stats=True)
        s2, h2 = self._inorm(c2, st2)
        N, H, Wd, C = c2.shape
        out = torch.empty_like(c2)
        if f'{p}.downsample.0.weight' in w:
            d, std = self.o.conv(x, w[f'{p}.downsample.0.weight'].view(C, 1, 1, -1), w[f'{p}.downsample.0.bias'],
                                 stride=stride, pad=0, ksize=1, stats=True, bounded=True)
            sd, hd = self._inorm(d, std)
            L.call('keep_gm_join', d, sd, hd, c2, s2, h2, out, N, H * Wd, C)
        else:
            L.call('keep_gm_join', x, None, None, c2, s2, h2, out, N, H * Wd, C)
        return out

    def _gm_layer(self, src, tgt, p, ffn, h8, w8, shift, kv_rot, n_img):
        """GM/transformer.py:148-187 with the window partition inside the attention kernel.
        ``tgt`` holds the target tokens in SOURCE image order; image i attends to image (i+kv_rot) % n_img of it."""
        w = self.w
        C = src.shape[-1]
        Ltok = h8 * w8
        wqkv = w[f'{p}.qkv.weight']
        o = torch.empty_like(src)
        if tgt is src:
            qkv = self.o.linear(src, wqkv, out_bf16=True, bounded=True, n_img=n_img)
            q, k, v = qkv, ops.offset(qkv, C), ops.offset(qkv, 2 * C)
            sq = skv = (Ltok * 3 * C, 3 * C, 0)
        else:
            q = self.o.linear(src, wqkv[:C], out_bf16=True, bounded=True, n_img=n_img)
            kv = self.o.linear(tgt, wqkv[C:], out_bf16=True, bounded=True, n_img=n_img)
            k, v = kv, ops.offset(kv, C)
            sq, skv = (Ltok * C, C, 0), (Ltok * 2 * C, 2 * C, 0)
        self.o.attention(q, k, v, o, B=n_img * 4, H=1, Lq=Ltok // 4, Lk=Ltok // 4, D=C, Dv=C, scale=1.0 / (C ** 0.5),
                      q_str=sq, k_str=skv, v_str=skv, o_str=(Ltok * C, C, 0), mode=2, img_h=h8, img_w=w8, ksplit=2,
                      shift=shift, kv_rot=kv_rot, n_img=n_img)
        n1 = (w[f'{p}.norm1.weight'], w[f'{p}.norm1.bias'], 1e-5)
        fuse = self.o.ln_fusable(w[f'{p}.merge.weight'], Ltok)      # LayerNorm in the GEMM's epilogue (keep_conv2d ln_gamma, ABI v16)
        if not ffn:
            if fuse:
                return self.o.linear(o, w[f'{p}.merge.weight'], bounded=True, n_img=n_img, ln=n1, residual=src)
            m = self.o.linear(o, w[f'{p}.merge.weight'], bounded=True, n_img=n_img)
            return ops.layernorm(m, n1[0], n1[1], res=src)
        if fuse:
            m = self.o.linear(o, w[f'{p}.merge.weight'], bounded=True, n_img=n_img, ln=n1)
        else:
            m = ops.layernorm(self.o.linear(o, w[f'{p}.merge.weight'], bounded=True, n_img=n_img), n1[0], n1[1])
        if (self.o.mma == L.MMA_X3 and C == 128 and FUSED_GM_FFN_X3 and self.o.x3_twin(w[f'{p}.mlp.0.weight']) is not None):
            # mlp.0 -> GELU -> mlp.2 -> norm2 -> + src as ONE launch: the [M, 8C] intermediate (10 GB at 16 clips) stays in registers
            return self.o.gm_ffn_x3(src, m, w[f'{p}.mlp.0.weight'], w[f'{p}.mlp.2.weight'], w[f'{p}.norm2.weight'], w[f'{p}.norm2.bias'], 1e-5)
        if self.o.mma == L.MMA_BF16 and C == 128 and FUSED_GM_MLP:
            m2 = self.o.gm_mlp(src, m, w[f'{p}.mlp.0.weight'], w[f'{p}.mlp.2.weight'])      # [M,8C] never leaves the CU
        else:
            if self.o.mma == L.MMA_X3:      # cat[src | m] folded into the GEMM: two K-concatenated inputs (keep_conv2d in2)
                hmid = self.o.linear(src, w[f'{p}.mlp.0.weight'], act=L.ACT_GELU, bounded=True, x2=m, n_img=n_img)
            else:
                hmid = self.o.linear(ops.concat2(src, m), w[f'{p}.mlp.0.weight'], act=L.ACT_GELU, bounded=True, n_img=n_img)
            if self.o.ln_fusable(w[f'{p}.mlp.2.weight'], Ltok):
                return self.o.linear(hmid, w[f'{p}.mlp.2.weight'], bounded=True, n_img=n_img, residual=src,
                                     ln=(w[f'{p}.norm2.weight'], w[f'{p}.norm2.bias'], 1e-5))
            m2 = self.o.linear(hmid, w[f'{p}.mlp.2.weight'], bounded=True, n_img=n_img)
        return ops.layernorm(m2, w[f'{p}.norm2.weight'], w[f'{p}.norm2.bias'], res=src)

    def _gm_backbone(self, img_nchw):
        """GMFlow CNN encoder (GM/backbone.py) on [N,3,H,W] frames in [-1,1] -> 1/8-resolution features [N,h8,w8,C].
        Per-image arithmetic only (InstanceNorm, convolutions), so a frame's features do not depend on its pair."""
        w = self.w
        pfx = 'flownet.model'
        ws2d = w.get(f'{pfx}.backbone.conv1.weight_s2d') if GM_S2D_CONV1 else None
        if ws2d is not None and img_nchw.shape[2] % 2 == 0 and img_nchw.shape[3] % 2 == 0:
            # the 7x7 stride-2 convolution as a 4x4 stride-1 convolution on the 2x2 space-to-depth image (16-channel rows: the MFMA
            # kernels take it; the 147-deep element-wise gather of the 3-channel form ran at 0.5 TB/s)
            img = ops.rgb_s2d(img_nchw.contiguous())                                       # [N,H/2,W/2,16] normalised
            f, fst = self.o.conv(img, ws2d, None, stride=1, pad=2, ksize=4, stats=True, bounded=True, out_hw=img.shape[1:3])
        else:
            img = ops.nchw_to_nhwc(img_nchw, mode=1)                                       # [N,H,W,3] normalised
            f, fst = self.o.conv(img, w[f'{pfx}.backbone.conv1.weight'], None, stride=2, pad=3, ksize=7, stats=True)
        s, hh = self._inorm(f, fst)
        x = torch.empty_like(f)
        L.call('keep_affine_act', f, s, hh, x, f.shape[0], f.shape[1] * f.shape[2], f.shape[3], L.ACT_RELU)
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            x = self._gm_resblock(x, f'{pfx}.backbone.layer{li}.0', stride)
            x = self._gm_resblock(x, f'{pfx}.backbone.layer{li}.1', 1)
        return self.o.linear(x, w[f'{pfx}.backbone.conv2.weight'], w[f'{pfx}.backbone.conv2.bias'], bounded=True)

    def _gmflow(self, im1, im2):
        """im1, im2 [P,3,H,W] NCHW in [-1,1] -> backward flow [P,H,W,2] (channels-last: (dx, dy))."""
        P = im1.shape[0]
        return self._gmflow_pairs(self._gm_backbone(torch.cat([im1, im2], dim=0)), P)

    def _gmflow_clip(self, x):
        """K1 for a batch of clips x [B,T,3,H,W]: flownet(x[:,1:], x[:,:-1]) (KA:976-986) -> [B*(T-1),H,W,2].
        The reference pushes every interior frame through the CNN encoder twice (once as the first image of pair t,
        once as the second image of pair t+1); here the encoder runs once per frame and the pair batch
        [x[:,1:] ; x[:,:-1]] is assembled from the B*T feature maps (same values, 47 % less encoder work at T=20)."""
        B, T = x.shape[:2]
        feat = self._gm_backbone(x.reshape(B * T, *x.shape[2:]))
        key = ('pairs', B, T, str(feat.device))
        if key not in self._const:                      # (a host -> device copy: must not happen inside a graph capture)
            first = [b * T + t + 1 for b in range(B) for t in range(T - 1)]
            second = [b * T + t for b in range(B) for t in range(T - 1)]
            self._const[key] = torch.tensor(first + second, device=feat.device, dtype=torch.long)
        if GM_DEDUP_L0:
            return self._gmflow_pairs(None, B * (T - 1), uniq=(feat, self._const[key]))
        return self._gmflow_pairs(feat.index_select(0, self._const[key]), B * (T - 1))

    def _gmflow_pairs(self, feat, P, uniq=None):
        """feat [2P,h8,w8,C]: features of the P first images followed by the P second images -> flow [P,H,W,2].
        ``uniq=(frame features [F,h8,w8,C], pair index [2P])``: the clip form.  The position table and the self-attention block of
        layer 0 see one image at a time and nothing of its pair, so they run once per FRAME (an interior frame sits in two pairs:
        47 % fewer images at T = 20) and their outputs are gathered into pair order -- the same values (per-image arithmetic, per-image
        plans), `test_gmflow_clip_layer0_runs_once_per_frame`."""
        w = self.w
        pfx = 'flownet.model'
        n_img, (_, h8, w8, C) = 2 * P, (feat if uniq is None else uniq[0]).shape
        Ltok = h8 * w8
        table, grid = self._gm_consts(h8, w8, (feat if uniq is None else uniq[0]).device)
        s_att0 = None
        if uniq is None:
            c0 = ops.add_bcast(feat, table).view(n_img * Ltok, C)
        else:
            fu, sel = uniq
            F_ = fu.shape[0]
            cu = ops.add_bcast(fu, table).view(F_ * Ltok, C)
            su = self._gm_layer(cu, cu, f'{pfx}.transformer.layers.0.self_attn', False, h8, w8, 0, 0, F_)
            c0 = cu.view(F_, Ltok * C).index_select(0, sel).view(n_img * Ltok, C)
            s_att0 = su.view(F_, Ltok * C).index_select(0, sel).view(n_img * Ltok, C)
        wsz = h8 // 2
        for i in range(GMFLOW['num_layers']):
            shift = wsz // 2 if i % 2 == 1 else 0
            lp = f'{pfx}.transformer.layers.{i}'
            # the cross-attention target is the swapped INPUT of this block (concat1 is refreshed only after
            # the block, GM/transformer.py:308-317), not the output of its self-attention
            s_att = s_att0 if (i == 0 and s_att0 is not None) else self._gm_layer(c0, c0, f'{lp}.self_attn', False, h8, w8, shift, 0, n_img)
            c0 = self._gm_layer(s_att, c0, f'{lp}.cross_attn_ffn', True, h8, w8, shift, P, n_img)
        f0 = c0[:P * Ltok]
        f1 = c0[P * Ltok:]
        sF = (Ltok * C, C, 0)
        # global correlation soft-argmax (GM/matching.py:15-34): V = pixel grid, shared by all pairs
        corr = ops.empty((P * Ltok, 2), f0)
        self.o.attention(f0, f1, grid, corr, B=P, H=1, Lq=Ltok, Lk=Ltok, D=C, Dv=2, scale=1.0 / (C ** 0.5),
                      q_str=sF, k_str=sF, v_str=(0, 2, 0), o_str=(Ltok * 2, 2, 0))
        flow = ops.add_bcast(corr, grid, alpha=-1.0)
        # flow propagation (GM/transformer.py:363-372): k projected from the projected q
        fp = f'{pfx}.feature_flow_attn'
        q = self.o.linear(f0, w[f'{fp}.q_proj.weight'], w[f'{fp}.q_proj.bias'], bounded=True, n_img=P)
        k = self.o.linear(q, w[f'{fp}.k_proj.weight'], w[f'{fp}.k_proj.bias'], bounded=True, n_img=P)
        flow2 = ops.empty((P * Ltok, 2), f0)
        self.o.attention(q, k, flow, flow2, B=P, H=1, Lq=Ltok, Lk=Ltok, D=C, Dv=2, scale=1.0 / (C ** 0.5),
                      q_str=sF, k_str=sF, v_str=(Ltok * 2, 2, 0), o_str=(Ltok * 2, 2, 0))
        # convex upsampling (GM/gmflow.py:75-88)
        cat = ops.concat2(flow2, f0, pad_to=16)                  # [.., 130 -> 144]; upsampler.0.weight is packed to match
        cat = cat.view(P, h8, w8, cat.shape[-1])
        m = self.o.conv(cat, w[f'{pfx}.upsampler.0.weight'], w[f'{pfx}.upsampler.0.bias'], act=L.ACT_RELU)
        mask = self.o.linear(m, w[f'{pfx}.upsampler.2.weight'], w[f'{pfx}.upsampler.2.bias'], bounded=True)
        k8 = GMFLOW['upsample_factor']
        up = ops.empty((P, h8 * k8, w8 * k8, 2), f0)
        L.call('keep_convex_upsample', mask, flow2, up, P, h8, w8, k8)
        return up

    # ------------------------------------------------------------------ KEEP.forward
    @torch.no_grad()
    def __call__(self, x, need_upscale=False, force_indices=None, return_aux=False, force_flows=None, _defer_check=False):
        if self.w is None:
            raise RuntimeError("KeepNet: weights are not on a device (load_state_dict + .to('cuda') first)")
        if x.dim() != 5 or x.shape[2] != 3:
            raise ValueError(f"expected [B,T,3,H,W], got {tuple(x.shape)}")
        cfg = self.cfg
        x = x.to(device=self.device, dtype=torch.float32)
        if need_upscale:
            # KA:1020-1023: x4 bilinear pre-upscale (never requested by the processor, which always passes False)
            Bn, Tn, _, h0, w0 = x.shape
            xs = x.contiguous()
            x = torch.empty((Bn, Tn, 3, 4 * h0, 4 * w0), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                L.call('keep_bilinear_upscale', xs, x, Bn * Tn * 3, h0, w0, 4)
        x = x.contiguous()
        B, T, _, H, Wd = x.shape
        if H % 32 or Wd % 32:
            raise ValueError("H and W must be multiples of 32")
        with torch.cuda.device(self.device):
            self._activate_precision()
            plain = force_indices is None and force_flows is None and not return_aux and self.o.profile is None
            if plain and (self.graph_mode == '1' or (self.graph_mode == 'auto' and B <= GRAPH_MAX_CLIPS)) and not ops.DEBUG_SYNC:
                res = self._forward_graphed(x, B, T, H, Wd)
            else:
                res = self._forward(x, B, T, H, Wd, force_indices, return_aux, force_flows)
            if _defer_check or self.precision != 'x3' or not CHECK_X3_RANGE:
                return res
            bits = self._status_bits()
            if bits == 0:
                return res
            del res                                  # the x3 result goes back to the allocator before the f32 re-run
            return self._checked(None, x, B, T, H, Wd, force_indices, return_aux, force_flows, bits=bits)

    def _status_bits(self):
        """This forward's status word (KEEP_STATUS_*): one 4-byte device -> host read."""
        return int(self.o.status.item())

    def _checked(self, res, x, B, T, H, Wd, force_indices=None, return_aux=False, force_flows=None, bits=None):
        """x3 policy: fp16 halves top out at 65504.  Raw-stream operands are range-probed, `bounded` ones are not; an operand
        beyond the range turns into NaN in everything it touches and every op between there and the output propagates it --
        including the discrete ones: the arg-max raises KEEP_STATUS_NONFINITE_LOGITS (it would otherwise select code 0 and the
        generator would paint a finite, wrong frame), ReLU / flow_warp keep NaN, and the last kernel of the forward scans the
        output (KEEP_STATUS_NONFINITE_TENSOR).  Any bit set -> the batch is re-run on the exact-f32 kernels: never a quietly
        wrong frame.  The check is ONE int32 read (the forward's status word), not a host-side reduction over the output."""
        if self.precision != 'x3' or not CHECK_X3_RANGE:
            return res
        if bits is None:
            bits = self._status_bits()
        if bits == 0:
            return res
        import logging
        logging.getLogger('ComfyUI-KEEP').warning(
            "x3 precision policy left the fp16 operand range on this batch (status %d); re-running it on the f32 kernels", bits)
        self.x3_fallbacks += 1
        self.precision = 'fp32'
        try:
            self._activate_precision()
            res = None                               # (callers drop their reference to the x3 result before calling: the re-run needs the room)
            if not return_aux and force_indices is None and force_flows is None:
                # the exact-f32 policy holds 0.36 GB per frame against 0.22: a call sized for x3 (by free HBM, up to 48 clips) re-runs in
                # parts sized by what is free NOW under the f32 footprint (per-image plans: same bits as one call)
                part = max(1, min(16, self.clips_per_call(T, H, Wd)))
                if B > part:
                    return torch.cat([self._forward(x[b0:b0 + part].contiguous(), min(part, B - b0), T, H, Wd, None, False)
                                      for b0 in range(0, B, part)], 0)
            return self._forward(x, B, T, H, Wd, force_indices, return_aux, force_flows)
        finally:
            self.precision = 'x3'
            self._activate_precision()

    def _forward_graphed(self, x, B, T, H, Wd):
        """The same kernel sequence as ``_forward``, captured once per (shape, policy) into a hipGraph and replayed: at
        B = 1 a clip is ~9 k launches averaging a few microseconds of GPU work each, i.e. bound by the host's launch
        rate; a replay submits them in one call.  Every kernel is stream-ordered, allocation-free and deterministic, and
        the host code between launches only computes shapes, so the replay is bit-identical to the eager run."""
        self.o.ensure_arena(self.device)
        key = (B, T, H, Wd, self.precision, self._dev_blob.data_ptr(), self.o.arena_generation, self.o.flags, self.o.attn_flags,
               self.o.plan_ref_images)
        ent = self._graphs.get(key)
        if ent is None:
            # First occurrence of a key: run eagerly (it is also the warm-up: weight twins, constants, plans, allocator).
            # Capture on the SECOND one -- a capture costs a host-side pass over ~9 k launches and holds a private activation
            # pool; a shape that shows up once per node call (a remainder clip) never pays for it.  KEEP_AMD_GRAPH=1 captures
            # at once (after one warm pass).
            seen = self._graph_seen.get(key, 0)
            self._graph_seen[key] = seen + 1
            if seen == 0:
                out = self._forward(x, B, T, H, Wd, None, False, None)
                if self.graph_mode != '1':
                    return out
            torch.cuda.synchronize()
            if len(self._graphs) >= GRAPH_CACHE:
                self._graphs.pop(next(iter(self._graphs)))
            static_x = x.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self._forward(static_x, B, T, H, Wd, None, False, None)
            ent = self._graphs[key] = (g, static_x, static_out)
        g, static_x, static_out = ent
        static_x.copy_(x)
        g.replay()
        return static_out.clone()

    def _frame(self, t5, i):
        """[B,T,...] -> frame i as a contiguous [B,...] (free view when B == 1)."""
        return t5[:, i].contiguous()

    def _forward(self, x, B, T, H, Wd, force_indices, return_aux, force_flows=None):
        cfg = self.cfg
        self.o.begin_forward(self.device)
        self._aux_top1 = []
        # K1: flows for all T-1 pairs (KA:976-986): flownet(x[:,1:], x[:,:-1])
        flows = None
        # Few clips in flight (the literal configs[1]: ONE): the frame recurrence below is a chain of ~8 000 small launches that
        # leaves most of the chip idle, while K1 (GMFlow) and K3 (Kalman gains) are wide, batched and needed only from frame 1 on
        # (frame i reads the flow of pair i - 1 and gains[i]).  They run on a SECOND stream, GMFlow in chunks of pairs, each chunk
        # fenced by an event the frame loop waits for just before its first use -- the batched stages fill the CUs the chain leaves
        # empty.  Same kernels on the same data (per-image arithmetic and plans: chunking the pair batch changes no bit), so the
        # result is bit-identical to the one-stream order; hipGraph capture records the fork / join.
        overlap = (STREAM_OVERLAP_MAX_CLIPS >= B and T > 2 and force_flows is None and not ops.DEBUG_SYNC and self.o.profile is None)
        flow_parts, ev_gain, side, main = None, None, None, None
        if force_flows is not None:      # parity tests: inject the oracle's flows [B,T-1,2,H,W] (isolates GMFlow drift)
            flows = force_flows.to(device=self.device, dtype=torch.float32).permute(0, 1, 3, 4, 2).contiguous()
        elif T > 1 and not overlap:
            with _Range('K1 gmflow'):
                flows = self._gmflow_clip(x)
            flows = flows.view(B, T - 1, H, Wd, 2)
        elif T > 1:
            main = torch.cuda.current_stream()
            side = self._side_stream = getattr(self, '_side_stream', None) or torch.cuda.Stream(device=self.device)
            side.wait_stream(main)                                       # fork (also orders the side pool's reuse behind the last forward)
            cuts = [0] + list(range(min(STREAM_OVERLAP_FIRST, T - 1), T - 1, STREAM_OVERLAP_CHUNK)) + [T - 1]
            cuts = sorted(set(cuts))
            flow_parts = []                                              # (first pair, one past the last, flows [B,n,H,W,2], event)

            def flow_chunk(a, b):
                with torch.cuda.stream(side), _Range('K1 gmflow (side stream)'):
                    f = self._gmflow_clip(x[:, a:b + 1]).view(B, b - a, H, Wd, 2)
                    ev = torch.cuda.Event()
                    ev.record(side)
                flow_parts.append((a, b, f, ev))
            flow_chunk(cuts[0], cuts[1])                                 # the first pairs: under K2 and frame 0
        # K2: LQ encoder over all B*T frames, stash CFT taps
        xn = ops.nchw_to_nhwc(x.view(B * T, 3, H, Wd))
        taps = [FUSE_ENCODER_BLOCK[s] for s in cfg['cft_list']]
        tap_st = {}
        with _Range('K2 lq_encoder'):
            z, feats = self._vq_stack(xn, 'encoder', encoder_blocks(cfg), taps, tap_stats=tap_st)
            # the encoder half of every CFT block's first convolution + shortcut: all B*T frames in one launch each (they read the LQ taps only)
            cft_pre = {k: tuple(t.view(B, T, *t.shape[1:]) for t in self._cft_enc_part(v, f'cft.{k}', tap_st.get(k)))
                       for k, v in feats.items() if k in cfg['cft_list'] and self._cft_splittable(f'cft.{k}', v.shape[-1])}
        enc_feat = {k: v.view(B, T, *v.shape[1:]) for k, v in feats.items()}
        zc = z.view(B, T, *z.shape[1:])
        # K3: Kalman gains over the whole clip.  They only enter frames i >= 1 (KA:1067-1070), so a T = 1 "clip" (the
        # single-image fast path: frame 0 depends on neither the flows nor the gains nor the other frames) skips them.
        if side is not None:
            ev_z = torch.cuda.Event()
            ev_z.record(main)
            with torch.cuda.stream(side), _Range('K3 kalman_gain (side stream)'):
                side.wait_event(ev_z)
                gains = self._kalman_gain(z, B, T).view(B, T, -1)
                ev_gain = torch.cuda.Event()
                ev_gain.record(side)
            for a, b in zip(cuts[1:-1], cuts[2:]):                       # the remaining pairs, behind the gains, under the frame loop
                flow_chunk(a, b)
        else:
            with _Range('K3 kalman_gain'):
                gains = self._kalman_gain(z, B, T).view(B, T, -1) if T > 1 else None
        cft_at = {FUSE_GENERATOR_BLOCK[s]: s for s in cfg['cft_list']}
        cfa_at = {FUSE_GENERATOR_BLOCK[s]: s for s in cfg['cfa_list']}
        gblocks = generator_blocks(cfg)
        out_nhwc = ops.empty((B, T, H, Wd, 3), x)
        idx_all, margin_all = [], []
        cross_prev, cross_prev_amax = {}, {}
        prev_out = None
        fi = None
        if force_indices is not None:
            fi = force_indices.to(device=self.device, dtype=torch.int32).contiguous()
        for i in range(T):
            z_i = self._frame(zc, i)
            if i == 0:
                z_hat = z_i
            else:                                                        # K4 (KA:1067-1070)
                with _Range('K4 warp + hq_encoder + kalman_update'):
                    warped = torch.empty_like(prev_out)
                    if flow_parts is not None:                           # two-stream order: this pair's chunk (and, once, the gains)
                        a, b, fpart, ev = next(p for p in flow_parts if p[0] <= i - 1 < p[1])
                        if i - 1 == a:
                            main.wait_event(ev)
                        if i == 1:
                            main.wait_event(ev_gain)
                        flow_i = self._frame(fpart, i - 1 - a)
                    else:
                        flow_i = self._frame(flows, i - 1)
                    L.call('keep_flow_warp', prev_out, flow_i, warped, B, H, Wd, 3)
                    z_prime, _ = self._vq_stack(warped, 'hq_encoder', encoder_blocks(cfg))
                    z_hat = torch.empty_like(z_i)
                    L.call('keep_kalman_update', z_i, z_prime, self._frame(gains, i), z_hat, B,
                           z_i.shape[1] * z_i.shape[2], z_i.shape[3])
            with _Range('K5-K6 code transformer + argmax'):
                quant, idx, margin = self._predict_codes(                # K5, K6
                    z_hat, None if fi is None else self._frame(fi, i).view(-1), return_aux)
            idx_all.append(idx)
            margin_all.append(margin)

            def hook(j, y, yst, i=i):                                    # K7 taps (KA:1104-1121)
                if j in cft_at:
                    s = cft_at[j]
                    pre = None if s not in cft_pre else tuple(self._frame(t, i) for t in cft_pre[s])
                    y, yst = self._cft(self._frame(enc_feat[s], i), y, f'cft.{s}', pre, yst)
                if j in cfa_at:
                    s = cfa_at[j]
                    y_amax = None if yst is None else yst.amax                 # (frame 0: the producing convolution's fused max|out|)
                    if i > 0:
                        (y, y_amax), yst = self._cfa(y, cross_prev[s], f'cfa.{s}', y_amax, cross_prev_amax.get(s), want_amax=True), None
                    cross_prev[s], cross_prev_amax[s] = y, y_amax
                return y, yst

            with _Range('K7 generator + CFT + CFA'):
                y, _ = self._vq_stack(quant, 'generator', gblocks, hook=hook)
            prev_out = y
            out_nhwc[:, i].copy_(y)
        if side is not None:
            main.wait_stream(side)                                       # join (every chunk has been waited for; this closes a capture)
            if return_aux:
                flows = torch.cat([p[2] for p in flow_parts], dim=1)
        out = ops.nhwc_to_nchw(out_nhwc.view(B * T, H, Wd, 3)).view(B, T, 3, H, Wd)
        if self.precision == 'x3' and CHECK_X3_RANGE:
            L.call('keep_nonfinite_flag', out, out.numel(), self.o.status)
        if return_aux:
            aux = {'indices': torch.stack(idx_all, 1), 'margins': torch.stack(margin_all, 1), 'gains': gains,
                   'flows': flows, 'z_codes': zc, 'logit_top1': torch.stack(self._aux_top1, 1)}
            self.last_aux = aux
            return out, aux
        return out

    # ------------------------------------------------------------------ independent clips (hot loop #1)
    def clips_per_call(self, T, H=512, Wd=512):
        """How many equal-length clips ride the batch axis of one net call.  Every batched stage (LQ encoder, GMFlow,
        Kalman gain) holds all B*T frames at once, so B is bounded by free HBM: ~0.22 GB per 512x512 frame at fp32
        storage under the x3 policy, 0.35 under exact f32 (measured: B=16, T=20 peaks at 70.5 / 110 GB).  ``KEEP_AMD_MAX_CLIPS``
        caps it (default 48 = what 288 GB hold under x3: 16 / 32 / 48 clips per call run at 279.6 / 287.1 / 290.3 frames/s -- the
        16 x 16 ... 64 x 64 stages and the token GEMMs fill the chip better; results do not depend on the choice: per-image
        plans); the node's ``max_clip_length`` therefore still bounds memory: a longer clip means fewer clips per call, never a
        bigger footprint."""
        cap = int(os.environ.get('KEEP_AMD_MAX_CLIPS', '48'))
        per_frame = {'bf16': 0.17e9, 'fp32': 0.36e9}.get(self.precision, 0.23e9) * (H * Wd) / (512.0 * 512.0)
        try:
            free, _ = torch.cuda.mem_get_info(self.device)
            free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)   # cached blocks are reusable
        except Exception:
            free = 64e9
        return max(1, min(cap, int(0.8 * free / (per_frame * max(T, 1)))))

    def run_clips(self, clips, need_upscale=False, max_b=None):
        """list of [1,T_i,3,H,W] -> list of restored clips.  Clips share no state (KA:1050,1064,1113), so equal-length
        clips are stacked on the batch axis (as many as free HBM allows, ``clips_per_call``).  A clip's result does not
        depend on its batch-mates AT ALL under the parity policies: kernel tiles, split-K factors and statistics partitions
        follow the per-image geometry and a fixed reference batch (``keep_conv2d_plan``, DESIGN.md section 6 "batch
        invariance"), so batched == sequential bit for bit (tests/test_gpu_net.py::test_batched_clips_equal_sequential,
        ``torch.equal``)."""
        order = {}
        for n, c in enumerate(clips):
            order.setdefault((c.shape[1], c.shape[3], c.shape[4]), []).append(n)
        outs = [None] * len(clips)
        for (T, H, Wd), ids in order.items():
            b = self.clips_per_call(T, H, Wd) if max_b is None else max_b
            for s in range(0, len(ids), b):
                grp = ids[s:s + b]
                res = self(torch.cat([clips[n] for n in grp], dim=0), need_upscale=need_upscale)
                for k, n in enumerate(grp):
                    outs[n] = res[k:k + 1]
        return outs

    # ------------------------------------------------------------------ device-side pre/post
    def run_clips_u8(self, clips_u8, max_b=None, gather='root', sink=None, parse=False):
        """list of uint8 BGR crops [T_i,H,W,3] (host or device) -> list of restored uint8 BGR [T_i,H,W,3] on the host.

        Replaces the per-frame host conversions either side of the clip loop -- ``img2tensor(face/255., bgr2rgb) +
        normalize(0.5, 0.5)`` (keep_processor.py:258-259) and ``tensor2img(rgb2bgr, min_max=(-1,1))``
        (keep_processor.py:272-273, img_util.py:66-90) -- with ``keep_img2tensor`` / ``keep_tensor2img`` on the GPU, so
        only uint8 crosses PCIe (4x fewer bytes each way).  Bit-identical to the host converters.  Clips are converted,
        restored and converted back one batch group at a time (nothing but uint8 outlives a group); the transfers of
        neighbouring groups run on a second stream under the current group's forward (``_run_clips_u8_local``).

        With an initialised ``torch.distributed`` group of more than one rank (one process per GPU, engine/dist.py) the
        clips are sharded round-robin over the ranks -- no data-path collective, clips share no state -- and the restored
        uint8 clips are collected by clip index with ONE fixed-size uint8 tensor gather: ``gather='root'`` (default) rank 0
        returns the full list and every other rank None (the paste-back that follows runs once, on rank 0); ``'all'`` every
        rank returns the full list; ``'none'`` each rank returns its own clips as {clip index: tensor}.

        ``sink(ids, crops, classes)`` (single process or worker pool; not under a torch.distributed group): instead of collecting
        every restored clip on the host, each finished batch group is handed over as soon as its range check has passed --
        ``ids``: the clip indices, ``crops``: the matching list of restored uint8 [T,H,W,3] tensors, ON THIS PROCESS'S GPU for the
        groups it ran itself (no download at all) and in host memory for a pool worker's, ``classes``: None, or the ParseNet class
        maps uint8 [T,H,W] a worker computed for its crops (``GpuPool.set_parser``).  The call then returns None.  The processor's
        sequence path pastes frames from it while the next group is being restored (modules/keep_processor.py)."""
        if self.w is None:
            raise RuntimeError("KeepNet: weights are not on a device (load_state_dict + .to('cuda') first)")
        from . import dist as kdist
        clips_u8 = [c if isinstance(c, torch.Tensor) else _FrameList(c) for c in clips_u8]
        for c in clips_u8:
            if len(c.shape) != 4 or c.shape[-1] != 3 or c.dtype != torch.uint8:
                raise ValueError(f"expected uint8 [T,H,W,3], got {c.dtype} {tuple(c.shape)}")
        if self.pool is not None and self.shard_across_ranks and len(clips_u8) > 1:
            # single-process product (a ComfyUI node): this process is the root of a worker pool, one worker per additional GPU
            return self.pool.run(self, clips_u8, max_b, sink=sink, parse=bool(parse) and sink is not None)
        grouped = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if sink is not None and grouped and self.shard_across_ranks:
            raise ValueError("run_clips_u8(sink=...) is for one process (with or without the worker pool), not a torch.distributed job")
        if not self.shard_across_ranks or sink is not None:      # per-rank workloads (BASELINE configs[4]: one video per GPU): nothing to exchange
            local = self._run_clips_u8_local(dict(enumerate(clips_u8)), max_b, sink=sink)
            return None if sink is not None else [torch.from_numpy(local[i]) for i in range(len(clips_u8))]
        res = kdist.sharded_map(clips_u8, lambda mine: self._run_clips_u8_local(mine, max_b), gather,
                                shapes=[tuple(c.shape) for c in clips_u8])
        if res is None or isinstance(res, dict):
            return res
        return [torch.from_numpy(r) if not isinstance(r, torch.Tensor) else r for r in res]

    def _pinned_staging(self, shape, slot):
        """Pinned host staging for the uploads, cached per (shape, pipeline slot): hipHostMalloc of a 250 MB group costs tens of
        milliseconds, which a 15-clip call would pay in front of its only forward.  Two slots: group g+1 is staged while
        group g's upload may still be in flight."""
        key = (tuple(shape), slot)
        buf = self._pinned_in.get(key)
        if buf is None:
            if len(self._pinned_in) >= 4:
                self._pinned_in.clear()
            buf = self._pinned_in[key] = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
        return buf

    def _run_clips_u8_local(self, mine, max_b=None, sink=None):
        """{clip index: uint8 [T,H,W,3]} -> {clip index: restored uint8 numpy [T,H,W,3]} on this rank's GPU.

        Three-stage pipeline over the batch groups, two HIP streams: while group g runs its forward on the compute stream,
        the copy stream uploads group g+1 (pinned staging -> device uint8) and downloads group g-1's restored uint8 into
        pinned memory.  The x3 range check of a group (one int32, ``_checked``) is read after its download has been queued,
        i.e. the host never waits inside a forward; a flagged group is re-run on the f32 kernels before it is handed back.
        Clips that already live on this GPU (device tensors) skip the staging and the upload.  With ``sink`` nothing is downloaded:
        a finished (range-checked) group is handed to ``sink(ids, [uint8 device tensors], None)`` and {} is returned."""
        order = {}
        for n, c in mine.items():
            order.setdefault(tuple(c.shape[:3]), []).append(n)
        groups = []
        for (T, H, Wd), ids in order.items():
            b = self.clips_per_call(T, H, Wd) if max_b is None else max_b
            if sink is not None and max_b is None:
                b = min(b, 16)      # streamed paste-back: several groups, so that the paste of group k runs under the forward of group k + 1
            groups += [(T, H, Wd, ids[s:s + b]) for s in range(0, len(ids), b)]
        local = {}
        if not groups:
            return local
        with torch.cuda.device(self.device):
            comp = torch.cuda.current_stream()
            io = self._io_stream = getattr(self, '_io_stream', None) or torch.cuda.Stream(device=self.device)

            slot_ev = {}                                        # staging slot -> event of the H2D copy that last read it

            def on_device(c):
                return isinstance(c, torch.Tensor) and c.device == self.device

            def upload(gi):
                T, H, Wd, grp = groups[gi]
                if all(on_device(mine[n]) for n in grp):         # crops warped on this GPU (engine/paste.py:crop_faces): no PCIe round trip
                    dev = torch.stack([mine[n] for n in grp]) if len(grp) > 1 else mine[grp[0]].unsqueeze(0).contiguous()
                    ev = torch.cuda.Event()
                    ev.record(comp)
                    return None, dev, ev
                if (gi & 1) in slot_ev:                         # group gi - 2 was copied from this pinned buffer: the CPU must not
                    slot_ev[gi & 1].synchronize()               # overwrite it before that copy has finished (satisfied in steady state)
                host = self._pinned_staging((len(grp), T, H, Wd, 3), gi & 1)
                for k, n in enumerate(grp):
                    if isinstance(mine[n], _FrameList):         # T separate [H,W,3] crops: one copy each, straight into pinned
                        for t, fr in enumerate(mine[n].frames):
                            host[k, t].copy_(torch.from_numpy(fr))
                    else:
                        host[k].copy_(mine[n])                  # (a device-resident clip is copied by the same call)
                with torch.cuda.stream(io):
                    dev = host.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(io)
                slot_ev[gi & 1] = ev
                return host, dev, ev

            def finish(job):
                """download queued -> wait for it, check the group's status word, hand the frames back"""
                gi, x, r8_host, st_host, ev_done = job
                ev_done.synchronize()
                T, H, Wd, grp = groups[gi]
                bits = int(st_host.item())
                if bits and x is not None:
                    o = self._checked(None, x, len(grp), T, H, Wd, bits=bits)
                    y = ops.nchw_to_nhwc(o.view(len(grp) * T, 3, H, Wd))
                    r8 = torch.empty((len(grp) * T, H, Wd, 3), dtype=torch.uint8, device=self.device)
                    L.call('keep_tensor2img', y, r8, len(grp) * T * H * Wd)
                    if sink is not None:
                        r8_host = r8.view(len(grp), T, H, Wd, 3)
                        comp.synchronize()              # the sink reads the crops on ITS stream: the re-run (compute stream) must be complete
                    else:
                        r8_host = r8.view(len(grp), T, H, Wd, 3).cpu()
                if sink is not None:                    # r8_host is the DEVICE tensor here (nothing was downloaded)
                    sink(list(grp), [r8_host[k] for k in range(len(grp))], None)
                    return
                arr = r8_host.numpy()
                for k, n in enumerate(grp):
                    local[n] = arr[k]

            nxt = upload(0)
            pending = None
            for gi, (T, H, Wd, grp) in enumerate(groups):
                host, u8, ev_in = nxt
                nxt = upload(gi + 1) if gi + 1 < len(groups) else None     # flies under this group's forward
                comp.wait_event(ev_in)
                u8.record_stream(comp)
                f = torch.empty((len(grp) * T, H, Wd, 3), dtype=torch.float32, device=self.device)
                L.call('keep_img2tensor', u8, f, len(grp) * T * H * Wd)
                x = ops.nhwc_to_nchw(f).view(len(grp), T, 3, H, Wd)
                del f, u8
                o = self(x, _defer_check=True)
                y = ops.nchw_to_nhwc(o.view(len(grp) * T, 3, H, Wd))
                r8 = torch.empty((len(grp) * T, H, Wd, 3), dtype=torch.uint8, device=self.device)
                L.call('keep_tensor2img', y, r8, len(grp) * T * H * Wd)
                st_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                st_host.copy_(self.o.status, non_blocking=True)            # this forward's status word, stream-ordered
                ev_out = torch.cuda.Event()
                ev_out.record(comp)
                if sink is not None:                    # the crops stay on the GPU; only the status word comes back
                    r8_host, ev_done = r8.view(len(grp), T, H, Wd, 3), ev_out
                else:
                    r8_host = torch.empty((len(grp), T, H, Wd, 3), dtype=torch.uint8, pin_memory=True)
                    with torch.cuda.stream(io):
                        io.wait_event(ev_out)
                        r8_host.copy_(r8.view(len(grp), T, H, Wd, 3), non_blocking=True)
                        r8.record_stream(io)
                        ev_done = torch.cuda.Event()
                        ev_done.record(io)
                del o, y, r8
                if pending is not None:
                    finish(pending)          # group g-1: its download ran under group g's launches
                # x is kept only for the (rare) f32 re-run; precision 'x3' is the only policy that can ask for one
                pending = (gi, x if (self.precision == 'x3' and CHECK_X3_RANGE) else None, r8_host, st_host, ev_done)
            finish(pending)
        return local
