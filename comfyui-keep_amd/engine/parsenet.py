"""ParseNet (the face-parsing net behind ``FaceRestoreHelper.paste_faces_to_input_image(use_parse=True)``) on the
MI355X kernels -- SURVEY.md 8f-4, reference ``wm_facelib/parsing/parsenet.py:140-194`` as instantiated by
``init_parsing_model`` (``wm_facelib/parsing/__init__.py:24``: ``ParseNet(in_size=512, out_size=512)``).

The reference runs it once per restored face per frame on a batch of ONE (face_restoration_helper.py:418-424); here all
the faces handed over ride the batch axis of every kernel.  Structure (parsenet.py:163-194):

    encoder   ConvLayer(3 -> 64)                       then log2(in/32) ResidualBlock(scale='down')
    body      10 x ResidualBlock(256)                   x = feat + body(feat)
    decoder   log2(out/32) ResidualBlock(scale='up')
    out_mask_conv ConvLayer(64 -> 19)                   (out_img_conv is never read by the helper: not computed)

Every ConvLayer is ReflectionPad2d(1) -> Conv2d(3x3, stride 1 | 2) -> BatchNorm(eval) -> LeakyReLU(0.2) (parsenet.py:72-109):
reflection padding is ``keep_conv2d``'s ``pad_mode = KEEP_PAD_REFLECT``, the eval-mode BatchNorm is folded into the
convolution's weights and bias when the state dict is packed (w * gamma / sqrt(var + eps), beta - mean * gamma / sqrt(var +
eps): the same affine map, rounded once), the nearest x2 of the 'up' layers is the convolution's ``upsample`` gather, the
residual sum is the second convolution's epilogue.  19 class maps are padded to 32 output channels (zero weights) so that the
last layer runs on the LDS-halo kernel; ``keep_channel_argmax`` takes the arg-max over the 19 real ones.
"""
import math

import numpy as np
import torch

from . import hiplib as L
from . import ops
from .weights import pack_blob, views

BN_EPS = 1e-5          # nn.BatchNorm2d default (parsenet.py:22)
PARSING_CH = 19


def parsenet_spec(in_size=512, out_size=512, min_feat_size=32, base_ch=64, parsing_ch=PARSING_CH, res_depth=10, ch_range=(32, 256)):
    """Block list of ParseNet.__init__ (parsenet.py:155-187): [(name, kind, cin, cout)], kind in
    {'conv', 'down', 'none', 'up'} ('conv' = a bare ConvLayer with bias, no norm / activation)."""
    lo, hi = ch_range

    def clip(c):
        return max(lo, min(c, hi))
    min_feat_size = min(in_size, min_feat_size)
    down_steps = int(np.log2(in_size // min_feat_size))
    up_steps = int(np.log2(out_size // min_feat_size))
    blocks = [('encoder.0', 'conv', 3, base_ch)]
    head = base_ch
    for i in range(down_steps):
        blocks.append((f'encoder.{i + 1}', 'down', clip(head), clip(head * 2)))
        head *= 2
    for i in range(res_depth):
        blocks.append((f'body.{i}', 'none', clip(head), clip(head)))
    for i in range(up_steps):
        blocks.append((f'decoder.{i}', 'up', clip(head), clip(head // 2)))
        head //= 2
    blocks.append(('out_mask_conv', 'conv', clip(head), parsing_ch))
    return blocks


def parsenet_state_dict_spec(**kw):
    """name -> shape of the reference module's state dict (strict load), incl. the unused out_img_conv."""
    spec = {}

    def conv(p, cin, cout, bias):
        spec[f'{p}.conv2d.weight'] = (cout, cin, 3, 3)
        if bias:
            spec[f'{p}.conv2d.bias'] = (cout,)

    def bn(p, c):
        for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
            spec[f'{p}.norm.norm.{leaf}'] = (c,)
        spec[f'{p}.norm.norm.num_batches_tracked'] = ()
    blocks = parsenet_spec(**kw)
    for name, kind, cin, cout in blocks:
        if kind == 'conv':
            conv(name, cin, cout, True)
            continue
        if not (kind == 'none' and cin == cout):
            conv(f'{name}.shortcut_func', cin, cout, True)
        conv(f'{name}.conv1', cin, cout, False)
        bn(f'{name}.conv1', cout)
        conv(f'{name}.conv2', cout, cout, False)
        bn(f'{name}.conv2', cout)
    last_cin = blocks[-1][2]
    conv('out_img_conv', last_cin, 3, True)
    return spec


def _fold_bn(w, sd, p):
    """Conv2d(bias=False) + BatchNorm2d(eval) -> (w', b')."""
    g, b = sd[f'{p}.norm.norm.weight'].double(), sd[f'{p}.norm.norm.bias'].double()
    m, v = sd[f'{p}.norm.norm.running_mean'].double(), sd[f'{p}.norm.norm.running_var'].double()
    s = g / torch.sqrt(v + BN_EPS)
    return (w.double() * s.view(-1, 1, 1, 1)).float(), (b - m * s).float()


class ParseNetEngine:
    """``engine = ParseNetEngine(state_dict).to('cuda')``; ``engine.logits(x)`` / ``engine.classes(x)`` with x fp32 [N,3,S,S]
    in [-1, 1] (the helper's ``normalize(face/255, 0.5, 0.5)`` input, face_restoration_helper.py:420-421)."""

    def __init__(self, state_dict, in_size=512, out_size=512, precision='x3'):
        self.in_size, self.out_size = in_size, out_size
        self.blocks = parsenet_spec(in_size=in_size, out_size=out_size)
        spec = parsenet_state_dict_spec(in_size=in_size, out_size=out_size)
        sd = {k.replace('module.', ''): v for k, v in state_dict.items()}          # (parsing/__init__.py:43-46)
        missing = [k for k in spec if k not in sd]
        bad = [k for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
        if missing or bad:
            raise RuntimeError(f"ParseNetEngine: state dict does not match ParseNet(in_size={in_size}, out_size={out_size}): "
                               f"missing {missing[:4]}, shape mismatch {bad[:4]}")
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        t = {}

        def put(name, w, b):
            cout = w.shape[0]
            if cout % 32:                                  # 19 class maps -> 32 (zero rows): LDS-halo kernel geometry
                pad = 32 - cout % 32
                w = torch.cat([w, torch.zeros((pad,) + tuple(w.shape[1:]))], 0)
                b = torch.cat([b, torch.zeros(pad)], 0)
            t[f'{name}.weight'] = w.permute(0, 2, 3, 1).contiguous()            # [Cout,KH,KW,Cin]
            t[f'{name}.bias'] = b.contiguous()
        for name, kind, cin, cout in self.blocks:
            if kind == 'conv':
                put(name, sd[f'{name}.conv2d.weight'], sd[f'{name}.conv2d.bias'])
                continue
            if f'{name}.shortcut_func.conv2d.weight' in sd:
                put(f'{name}.shortcut', sd[f'{name}.shortcut_func.conv2d.weight'], sd[f'{name}.shortcut_func.conv2d.bias'])
            for c in ('conv1', 'conv2'):
                put(f'{name}.{c}', *_fold_bn(sd[f'{name}.{c}.conv2d.weight'], sd, f'{name}.{c}'))
        self._blob, self._index = pack_blob(t)
        self.precision = precision
        self.device = torch.device('cpu')
        self.w = None
        self.o = ops.Ops()

    def packed(self):
        """(packed fp32 blob, index, in_size, out_size, precision): what ``from_packed`` rebuilds the engine from in another process
        (engine/pool.py: the workers of the GPU pool parse the crops they restored)."""
        return self._blob, self._index, self.in_size, self.out_size, self.precision

    @classmethod
    def from_packed(cls, blob, index, in_size=512, out_size=512, precision='x3'):
        self = cls.__new__(cls)
        self.in_size, self.out_size = in_size, out_size
        self.blocks = parsenet_spec(in_size=in_size, out_size=out_size)
        self._blob, self._index = np.ascontiguousarray(blob), index
        self.precision = precision
        self.device = torch.device('cpu')
        self.w = None
        self.o = ops.Ops()
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            self.w, self._dev = None, None
            self.o.set_precision(self.o.mma)
            self.device = device
            return self
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        L.load(check_device=True)
        self.device = device
        self._dev = torch.from_numpy(self._blob).to(device)
        self.w = views(self._dev, self._index)
        if self.precision == 'x3':
            names = [n for n, (_, sh) in self._index.items() if len(sh) == 4 and sh[-1] % 16 == 0]
            bx, table = ops.make_x3_blob(self._dev, self._index, self.w, names)       # one power-of-two scale per tensor
            self.o.set_precision(L.MMA_X3, self._dev, None, bx, 1.0, x3_scales=table)
        else:
            self.o.set_precision(L.MMA_F32, self._dev, None)
        return self

    def _conv(self, x, name, x_amax=None, **kw):
        """-> (y, per-image max |y| from the launch's epilogue or None): the x3 range scale of the convolutions that read y, so that none
        of them probes its input (46 ``keep_absmax`` launches = 9 % of a 16-face call otherwise)."""
        y, st = self.o.conv(x, self.w[f'{name}.weight'], self.w[f'{name}.bias'], pad=1, ksize=3, reflect=True, stats='amax', x_amax=x_amax, **kw)
        return y, (None if st is None else st.amax)

    @torch.no_grad()
    def logits_nhwc(self, x_nhwc):
        """x [N,S,S,3] fp32 NHWC -> mask logits [N,S_out,S_out,32] (channels 19.. are zero)."""
        if self.w is None:
            raise RuntimeError("ParseNetEngine: call .to('cuda') first")
        with torch.cuda.device(self.device):
            self.o.begin_forward(self.device)
            x = x_nhwc.contiguous()
            xa = None                                                  # max |x| per image where a producer's epilogue supplied it
            body_in = None
            n_body = sum(1 for b in self.blocks if b[1] == 'none')
            seen_body = 0
            for name, kind, cin, cout in self.blocks:
                if kind == 'conv':
                    x, xa = self._conv(x, name, xa)
                    continue
                if kind == 'none' and seen_body == 0:
                    body_in = x                                        # feat (parsenet.py:190)
                if kind == 'down':       # parsenet.py:126-136: shortcut s2; conv1 s1 + BN + LReLU; conv2 s2 + BN; sum
                    idt, _ = self._conv(x, f'{name}.shortcut', xa, stride=2)
                    h, ha = self._conv(x, f'{name}.conv1', xa, act=L.ACT_LRELU02)
                    x, xa = self._conv(h, f'{name}.conv2', ha, stride=2, residual=idt)
                elif kind == 'up':       # shortcut and conv1 read the nearest x2 upsampling; conv2 at the new size
                    idt, _ = self._conv(x, f'{name}.shortcut', xa, upsample=True)
                    h, ha = self._conv(x, f'{name}.conv1', xa, upsample=True, act=L.ACT_LRELU02)
                    x, xa = self._conv(h, f'{name}.conv2', ha, residual=idt)
                else:
                    h, ha = self._conv(x, f'{name}.conv1', xa, act=L.ACT_LRELU02)
                    seen_body += 1
                    # the last body block also adds `feat` (x = feat + body(feat), parsenet.py:190): one more residual pass
                    x, xa = self._conv(h, f'{name}.conv2', ha, residual=x)
                    if seen_body == n_body:
                        x, xa = ops.add_bcast(x, body_in), None
            return x

    def logits(self, x_nchw):
        """Reference call shape: x [N,3,S,S] -> out_mask [N,19,S_out,S_out] (``face_parse(x)[0]``)."""
        x = x_nchw.to(device=self.device, dtype=torch.float32)
        y = self.logits_nhwc(ops.nchw_to_nhwc(x.contiguous()))
        return ops.nhwc_to_nchw(y)[:, :PARSING_CH]

    def classes(self, x_nhwc):
        """x [N,S,S,3] fp32 NHWC on the device -> uint8 class maps [N,S_out,S_out] (``out.argmax(dim=1)``, :424)."""
        y = self.logits_nhwc(x_nhwc)
        N, H, W, ld = y.shape
        out = torch.empty((N, H, W), dtype=torch.uint8, device=y.device)
        with torch.cuda.device(self.device):
            L.call('keep_channel_argmax', y, out, N * H * W, PARSING_CH, ld)
        return out


class EngineFaceParse:
    """Drop-in for ``FaceRestoreHelper.face_parse`` (an nn.Module called as ``face_parse(x)[0]``,
    face_restoration_helper.py:423): same call, the ParseNet runs on the HIP engine."""

    def __init__(self, engine):
        self.engine = engine

    @classmethod
    def from_module(cls, module, device=None, precision='x3'):
        """Take over the weights of a loaded reference ParseNet (sizes are read off its state dict).  ``device=None`` leaves the
        packed weights on the host until ``.to('cuda')`` (what KEEPModelPack.load_device() calls)."""
        sd = module.state_dict()
        n_down = sum(1 for k in sd if k.startswith('encoder.') and k.endswith('.shortcut_func.conv2d.weight'))
        n_up = sum(1 for k in sd if k.startswith('decoder.') and k.endswith('.shortcut_func.conv2d.weight'))
        eng = ParseNetEngine(sd, in_size=32 << n_down, out_size=32 << n_up, precision=precision)
        return cls(eng if device is None else eng.to(device))

    def __call__(self, x):
        return self.engine.logits(x), None

    def to(self, device):
        self.engine.to(device)
        return self

    def eval(self):
        return self


def synth_parsenet_state_dict(seed=0, **kw):
    """Deterministic synthetic ParseNet weights (engine/synth.py generator): conv weights ~ U with std 1/sqrt(fan_in), BatchNorm
    gamma 1 +- 0.1, beta / running_mean +- 0.1, running_var in [0.7, 1.3]."""
    from .synth import uniform_pm1
    out = {}
    for name, shape in parsenet_state_dict_spec(**kw).items():
        n = int(np.prod(shape)) if shape else 1
        u = uniform_pm1('parsenet.' + name, n, seed)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[name] = torch.tensor(100, dtype=torch.int64)
            continue
        if len(shape) == 4:
            v = u * (math.sqrt(3.0) / math.sqrt(shape[1] * 9))
        elif leaf == 'running_var':
            v = 1.0 + 0.3 * u
        elif leaf == 'weight':
            v = 1.0 + 0.1 * u
        else:
            v = 0.1 * u
        out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out
