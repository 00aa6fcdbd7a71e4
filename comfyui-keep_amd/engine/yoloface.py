"""YOLOv5-face detectors (``YOLOv5n`` / ``YOLOv5l`` of the node's detection_model list) on the MI355X kernels -- SURVEY.md 8f-4.

Reference: ``wm_facelib/detection/__init__.py:42-49`` (``YoloDetector(config_name=.../yolov5n.yaml | yolov5l.yaml)``),
``yolov5face/models/yolo.py:26-84`` (Detect), ``:87-131`` (Model: ``parse_model`` of the yaml, ``forward_once``),
``yolov5face/models/common.py:32-173`` (Conv, StemBlock, Bottleneck, C3, ShuffleV2Block, SPP, Concat) and the two yaml files
(layer lists restated below as ``YOLO_CFGS``).  This module is a drop-in for ``YoloDetector.detector`` -- ``model(images)[0]`` -> ``[N, anchors, 16]`` predictions in pixels of
the network input -- and, for the processor's batched pre-pass, ``yolo_detect_batch``: ``YoloDetector.detect_faces``
(face_detector.py:113-141) with the colour conversion, the letterbox and the float conversion in one kernel (``keep_yolo_letterbox_u8``),
the candidate selection of ``non_max_suppression_face`` (``keep_yolo_select``) and the suppression (``keep_retina_nms``) on the device:
uint8 frames go up, the handful of kept rows come back (a 1080p frame has 128 520 prediction rows = 8 MB).

How it maps onto the engine
  * Conv = Conv2d + BatchNorm2d(eval) + SiLU: one ``keep_conv2d`` with the folded weights and the ``KEEP_ACT_SILU`` epilogue;
  * Bottleneck's ``x + cv2(cv1(x))`` is the fused residual epilogue (the activation comes before the residual in the kernels);
  * C3 / SPP / Concat: the producers write their channel slices of ONE buffer (``out_ld``), pools read and write slices
    (``keep_maxpool2d``), ``nn.Upsample(2, nearest)`` + Concat is ``keep_slice_copy(up = 1)``;
  * ShuffleV2Block: 1x1 convolutions on a channel slice of the input (``in_off`` / ``cin``), ``keep_dwconv3x3`` for the depthwise
    3x3 + BatchNorm, ``keep_channel_shuffle2`` = ``channel_shuffle(cat(a, b), 2)`` in one pass;
  * StemBlock: 3 -> c 3x3 stride-2 convolution on the flattened-K kernel (Cin = 3), 2x2 stride-2 ceil-mode pool;
  * Detect: the level's 1x1 convolution (3 anchors x 16 outputs; NHWC = the reference's ``permute(0, 1, 3, 4, 2)``), then
    ``keep_yolo_decode`` writes the level's rows of the ``[N, anchors, 16]`` prediction tensor (sigmoid, grid / anchor decode).
"""
import math
import os

import numpy as np
import torch

from . import hiplib as L
from . import ops
from .weights import pack_blob, views

ANCHORS = ((4, 5, 8, 10, 13, 16), (23, 29, 43, 55, 73, 105), (146, 217, 231, 300, 335, 433))      # both yaml files
STRIDES = (8.0, 16.0, 32.0)                                                                       # Model.__init__, yolo.py:103-105
NO = 16                                                                                           # nc + 5 + 10, yolo.py:33
# (from, number, module, args) rows of yolov5n.yaml / yolov5l.yaml (depth_multiple = width_multiple = 1.0), Detect last
YOLO_CFGS = {
    'YOLOv5n': (
        (-1, 1, 'StemBlock', (32,)), (-1, 1, 'Shuffle', (128, 2)), (-1, 3, 'Shuffle', (128, 1)), (-1, 1, 'Shuffle', (256, 2)),
        (-1, 7, 'Shuffle', (256, 1)), (-1, 1, 'Shuffle', (512, 2)), (-1, 3, 'Shuffle', (512, 1)),
        (-1, 1, 'Conv', (128, 1, 1)), (-1, 1, 'Up', ()), ((-1, 4), 1, 'Concat', ()), (-1, 1, 'C3', (128, False)),
        (-1, 1, 'Conv', (128, 1, 1)), (-1, 1, 'Up', ()), ((-1, 2), 1, 'Concat', ()), (-1, 1, 'C3', (128, False)),
        (-1, 1, 'Conv', (128, 3, 2)), ((-1, 11), 1, 'Concat', ()), (-1, 1, 'C3', (128, False)),
        (-1, 1, 'Conv', (128, 3, 2)), ((-1, 7), 1, 'Concat', ()), (-1, 1, 'C3', (128, False)),
        ((14, 17, 20), 1, 'Detect', ())),
    'YOLOv5l': (
        (-1, 1, 'StemBlock', (64,)), (-1, 3, 'C3', (128, True)), (-1, 1, 'Conv', (256, 3, 2)), (-1, 9, 'C3', (256, True)),
        (-1, 1, 'Conv', (512, 3, 2)), (-1, 9, 'C3', (512, True)), (-1, 1, 'Conv', (1024, 3, 2)), (-1, 1, 'SPP', (1024, (3, 5, 7))),
        (-1, 3, 'C3', (1024, False)),
        (-1, 1, 'Conv', (512, 1, 1)), (-1, 1, 'Up', ()), ((-1, 5), 1, 'Concat', ()), (-1, 3, 'C3', (512, False)),
        (-1, 1, 'Conv', (256, 1, 1)), (-1, 1, 'Up', ()), ((-1, 3), 1, 'Concat', ()), (-1, 3, 'C3', (256, False)),
        (-1, 1, 'Conv', (256, 3, 2)), ((-1, 13), 1, 'Concat', ()), (-1, 3, 'C3', (512, False)),
        (-1, 1, 'Conv', (512, 3, 2)), ((-1, 9), 1, 'Concat', ()), (-1, 3, 'C3', (1024, False)),
        ((16, 19, 22), 1, 'Detect', ())),
}


def yolo_layers(name):
    """parse_model (yolo.py:175-235) on the restated yaml: [(index, from, kind, n, cin, cout, args)] with the channel bookkeeping."""
    rows, ch = [], []
    c2 = 3
    for i, (f, n, kind, args) in enumerate(YOLO_CFGS[name]):
        src = lambda j: 3 if (j == -1 and not ch) else ch[j]      # noqa: E731
        if kind in ('StemBlock', 'Shuffle', 'Conv', 'C3', 'SPP'):
            c1, c2 = src(f), args[0]
        elif kind == 'Concat':
            c1, c2 = None, sum(src(j) for j in f)
        elif kind == 'Detect':
            c1, c2 = tuple(ch[j] for j in f), None
        else:                                                     # Up
            c1 = c2 = src(f)
        rows.append((i, f, kind, n, c1, c2, args))
        ch.append(c2)
    return rows


def yolo_state_dict_spec(name):
    """name -> shape of Model(cfg).state_dict() (BatchNorm num_batches_tracked and Detect's anchor buffers included)."""
    spec = {}

    def conv(p, c1, c2, k, groups=1):          # Conv (common.py:32-45) / bare Conv2d + BatchNorm2d pairs
        spec[f'{p}.weight'] = (c2, c1 // groups, k, k)

    def bn(p, c):
        for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
            spec[f'{p}.{leaf}'] = (c,)
        spec[f'{p}.num_batches_tracked'] = ()

    def cb(p, c1, c2, k):
        conv(f'{p}.conv', c1, c2, k)
        bn(f'{p}.bn', c2)

    def c3(p, c1, c2, n):
        c_ = c2 // 2
        cb(f'{p}.cv1', c1, c_, 1)
        cb(f'{p}.cv2', c1, c_, 1)
        cb(f'{p}.cv3', 2 * c_, c2, 1)
        for k in range(n):
            cb(f'{p}.m.{k}.cv1', c_, c_, 1)
            cb(f'{p}.m.{k}.cv2', c_, c_, 3)

    def shuffle(p, inp, oup, stride):
        bf = oup // 2
        if stride > 1:
            conv(f'{p}.branch1.0', inp, inp, 3, groups=inp); bn(f'{p}.branch1.1', inp)
            conv(f'{p}.branch1.2', inp, bf, 1); bn(f'{p}.branch1.3', bf)
        conv(f'{p}.branch2.0', inp if stride > 1 else bf, bf, 1); bn(f'{p}.branch2.1', bf)
        conv(f'{p}.branch2.3', bf, bf, 3, groups=bf); bn(f'{p}.branch2.4', bf)
        conv(f'{p}.branch2.5', bf, bf, 1); bn(f'{p}.branch2.6', bf)

    for i, f, kind, n, c1, c2, args in yolo_layers(name):
        p = f'model.{i}'
        if kind == 'StemBlock':
            cb(f'{p}.stem_1', c1, c2, 3); cb(f'{p}.stem_2a', c2, c2 // 2, 1); cb(f'{p}.stem_2b', c2 // 2, c2, 3); cb(f'{p}.stem_3', 2 * c2, c2, 1)
        elif kind == 'Conv':
            cb(p, c1, c2, args[1])
        elif kind == 'C3':
            c3(p, c1, c2, n)                                       # (parse_model moves `number` into C3's n: one module)
        elif kind == 'SPP':
            cb(f'{p}.cv1', c1, c1 // 2, 1); cb(f'{p}.cv2', (c1 // 2) * (len(args[1]) + 1), c2, 1)
        elif kind == 'Shuffle':
            for k in range(n):                                     # nn.Sequential of n blocks with the SAME arguments
                shuffle(f'{p}.{k}' if n > 1 else p, c1, c2, args[1])
        elif kind == 'Detect':
            spec[f'{p}.anchors'] = (3, 3, 2)
            spec[f'{p}.anchor_grid'] = (3, 1, 3, 1, 1, 2)
            for k, c in enumerate(c1):
                spec[f'{p}.m.{k}.weight'] = (3 * NO, c, 1, 1)
                spec[f'{p}.m.{k}.bias'] = (3 * NO,)
    return spec


def config_of(state_dict):
    keys = set(state_dict)
    if 'model.0.stem_1.conv.weight' not in keys:
        raise RuntimeError("YoloFaceEngine: not a yolov5-face state dict")
    return 'YOLOv5n' if 'model.1.branch1.0.weight' in keys else 'YOLOv5l'


class YoloFaceEngine:
    """``bn_eps``: name of a BatchNorm -> its eps (the reference's Model keeps torch's 1e-5; checkpoints of the upstream trainer were
    made with 1e-3 -- ``EngineYoloModel.from_module`` reads the value off every module)."""

    def __init__(self, state_dict, precision='x3', bn_eps=None):
        self.name = config_of(state_dict)
        spec = yolo_state_dict_spec(self.name)
        missing = [k for k in spec if k not in state_dict]
        bad = [k for k in spec if k in state_dict and tuple(state_dict[k].shape) != tuple(spec[k])]
        if missing or bad:
            raise RuntimeError(f"YoloFaceEngine: not a {self.name} state dict: missing {missing[:4]}, shapes {bad[:4]}")
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if k in spec}
        eps = bn_eps or {}
        t = {}

        def fold(conv, bn):
            w = sd[f'{conv}.weight'].double()
            g, b = sd[f'{bn}.weight'].double(), sd[f'{bn}.bias'].double()
            m, v = sd[f'{bn}.running_mean'].double(), sd[f'{bn}.running_var'].double()
            s = g / torch.sqrt(v + eps.get(bn, 1e-5))
            return (w * s.view(-1, 1, 1, 1)).float(), (b - m * s).float()

        def put(name, w, b):
            if w.shape[1] == 1 and w.shape[0] > 1 and w.shape[2] == 3 and f'{name}' in self._dw:      # depthwise [C,1,3,3] -> tap-major [3,3,C]
                t[f'{name}.weight'], t[f'{name}.bias'] = w[:, 0].permute(1, 2, 0).contiguous(), b.contiguous()
                return
            w = w.permute(0, 2, 3, 1).contiguous()
            if w.shape[1] == 1 and w.shape[2] == 1:
                w = w.reshape(w.shape[0], w.shape[3])
            t[f'{name}.weight'], t[f'{name}.bias'] = w, b.contiguous()
        self._dw = set()
        self.layers = yolo_layers(self.name)
        for i, f, kind, n, c1, c2, args in self.layers:
            p = f'model.{i}'
            if kind == 'StemBlock':
                for s in ('stem_1', 'stem_2a', 'stem_2b', 'stem_3'):
                    put(f'{p}.{s}', *fold(f'{p}.{s}.conv', f'{p}.{s}.bn'))
            elif kind == 'Conv':
                put(p, *fold(f'{p}.conv', f'{p}.bn'))
            elif kind == 'C3':
                for s in ['cv1', 'cv2', 'cv3'] + [f'm.{k}.cv{q}' for k in range(n) for q in (1, 2)]:
                    put(f'{p}.{s}', *fold(f'{p}.{s}.conv', f'{p}.{s}.bn'))
            elif kind == 'SPP':
                for s in ('cv1', 'cv2'):
                    put(f'{p}.{s}', *fold(f'{p}.{s}.conv', f'{p}.{s}.bn'))
            elif kind == 'Shuffle':
                for k in range(n):
                    q = f'{p}.{k}' if n > 1 else p
                    pairs = [('branch2.0', 'branch2.1', False), ('branch2.3', 'branch2.4', True), ('branch2.5', 'branch2.6', False)]
                    if args[1] > 1:
                        pairs += [('branch1.0', 'branch1.1', True), ('branch1.2', 'branch1.3', False)]
                    for cv, bnn, dw in pairs:
                        if dw:
                            self._dw.add(f'{q}.{cv}')
                        put(f'{q}.{cv}', *fold(f'{q}.{cv}', f'{q}.{bnn}'))
            elif kind == 'Detect':
                for k in range(3):
                    put(f'{p}.m.{k}', sd[f'{p}.m.{k}.weight'], sd[f'{p}.m.{k}.bias'])
                # anchor_grid (pixels) is what the decode multiplies by (yolo.py:59-74); it travels with the weights
                t['anchor_grid'] = sd[f'{p}.anchor_grid'].reshape(3, 6).contiguous()
        self._blob, self._index = pack_blob(t)
        self.precision = precision
        self.device = torch.device('cpu')
        self.w = None
        self.o = ops.Ops()

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
            names = [n for n, (_, sh) in self._index.items() if len(sh) >= 2 and sh[-1] % 16 == 0 and n.endswith('.weight')
                     and n[:-7] not in self._dw and n != 'anchor_grid']
            bx, table = ops.make_x3_blob(self._dev, self._index, self.w, names)
            self.o.set_precision(L.MMA_X3, self._dev, None, bx, 1.0, x3_scales=table)
        else:
            self.o.set_precision(L.MMA_F32, self._dev, None)
        return self

    # ------------------------------------------------------------------ building blocks
    def _c(self, x, name, act=L.ACT_SILU, stride=1, **kw):
        """Conv (common.py:32-45) with the BatchNorm folded: k and padding k // 2 from the weight's shape."""
        w = self.w[f'{name}.weight']
        b = self.w[f'{name}.bias']
        if w.dim() == 2:
            return self.o.conv(x, w.view(w.shape[0], 1, 1, w.shape[1]), b, stride=stride, pad=0, ksize=1, act=act, **kw)
        return self.o.conv(x, w, b, stride=stride, pad=w.shape[1] // 2, ksize=w.shape[1], act=act, **kw)

    def _dwc(self, x, name, stride):
        N, H, W, C = x.shape
        out = ops.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, C), x)
        L.call('keep_dwconv3x3', x, self.w[f'{name}.weight'], self.w[f'{name}.bias'], out, N, H, W, C, stride, L.ACT_NONE)
        return out

    def _stem(self, x, p):
        s1 = self._c(x, f'{p}.stem_1', stride=2)
        a = self._c(s1, f'{p}.stem_2a')
        N, H, W, C = s1.shape
        Ho, Wo = -(-H // 2), -(-W // 2)                           # MaxPool2d(2, 2, ceil_mode=True); the 3x3 stride-2 conv gives the same size
        cat = ops.empty((N, Ho, Wo, 2 * C), s1)
        flat = cat.view(-1)
        self._c(a, f'{p}.stem_2b', stride=2, out=flat, out_ld=2 * C)
        L.call('keep_maxpool2d', s1, flat[C:], N, H, W, C, C, 2 * C, 2, 2, 0, Ho, Wo)
        return self._c(cat, f'{p}.stem_3')

    def _shuffle(self, x, p, stride):
        N, H, W, C = x.shape
        if stride == 1:
            half = C // 2
            y = self._c(x, f'{p}.branch2.0', cin=half, in_off=half)          # x2 = the second half of the channels
            y = self._dwc(y, f'{p}.branch2.3', 1)
            y = self._c(y, f'{p}.branch2.5')
            a, a_ld = x, C                                                      # x1 = the first half, untouched
        else:
            a = self._c(self._dwc(x, f'{p}.branch1.0', stride), f'{p}.branch1.2')
            y = self._c(x, f'{p}.branch2.0')
            y = self._dwc(y, f'{p}.branch2.3', stride)
            y = self._c(y, f'{p}.branch2.5')
            half = y.shape[3]
            a_ld = half
        out = ops.empty((N, y.shape[1], y.shape[2], 2 * half), y)
        L.call('keep_channel_shuffle2', a, y, out, N * y.shape[1] * y.shape[2], half, a_ld, half)
        return out

    def _c3(self, x, p, n, shortcut):
        N, H, W, _ = x.shape
        c_ = self.w[f'{p}.cv1.weight'].shape[0]
        cat = ops.empty((N, H, W, 2 * c_), x)
        flat = cat.view(-1)
        y = self._c(x, f'{p}.cv1')
        for k in range(n):                                          # Bottleneck(c_, c_, shortcut, e = 1.0), common.py:56-66
            h = self._c(y, f'{p}.m.{k}.cv1')
            kw = dict(residual=y) if shortcut else {}
            if k == n - 1:
                kw.update(out=flat, out_ld=2 * c_)
            y = self._c(h, f'{p}.m.{k}.cv2', **kw)
        self._c(x, f'{p}.cv2', out=flat[c_:], out_ld=2 * c_)
        return self._c(cat, f'{p}.cv3')

    def _spp(self, x, p, ks):
        N, H, W, _ = x.shape
        c_ = self.w[f'{p}.cv1.weight'].shape[0]
        C = c_ * (len(ks) + 1)
        cat = ops.empty((N, H, W, C), x)
        flat = cat.view(-1)
        self._c(x, f'{p}.cv1', out=flat, out_ld=C)
        for j, k in enumerate(ks):                                  # MaxPool2d(k, 1, k // 2) of cv1's output, common.py:160-163
            L.call('keep_maxpool2d', flat, flat[(j + 1) * c_:], N, H, W, c_, C, C, k, 1, k // 2, H, W)
        return self._c(cat, f'{p}.cv2')

    # ------------------------------------------------------------------ network
    @torch.no_grad()
    def forward_nhwc(self, x):
        """x [N,H,W,3] fp32 NHWC in [0, 1], RGB (what YoloDetector._preprocess makes), H and W multiples of 32 ->
        pred [N, anchors, 16]: ``Model.forward(x)[0]`` (yolo.py:81)."""
        if self.w is None:
            raise RuntimeError("YoloFaceEngine: call .to('cuda') first")
        N, H, W, _ = x.shape
        if H % 32 or W % 32:
            raise ValueError("YoloFaceEngine: the network input must be a multiple of the largest stride (32), as check_img_size / letterbox make it")
        with torch.cuda.device(self.device):
            self.o.begin_forward(self.device)
            ys = []
            y = x.contiguous()
            for i, f, kind, n, c1, c2, args in self.layers:
                p = f'model.{i}'
                if kind != 'Concat' and kind != 'Detect' and f != -1:
                    y = ys[f]
                if kind == 'StemBlock':
                    y = self._stem(y, p)
                elif kind == 'Conv':
                    y = self._c(y, p, stride=args[2])
                elif kind == 'C3':
                    y = self._c3(y, p, n, args[1])
                elif kind == 'SPP':
                    y = self._spp(y, p, args[1])
                elif kind == 'Shuffle':
                    for k in range(n):
                        y = self._shuffle(y, f'{p}.{k}' if n > 1 else p, args[1])
                elif kind == 'Up':
                    y = ('up', y)                                   # folded into the Concat that follows (always does, in both files)
                elif kind == 'Concat':
                    parts = [y if j == -1 else ys[j] for j in f]
                    shapes = [(q[1].shape[0], 2 * q[1].shape[1], 2 * q[1].shape[2], q[1].shape[3]) if isinstance(q, tuple) else q.shape
                              for q in parts]
                    C = sum(s[3] for s in shapes)
                    cat = ops.empty((shapes[0][0], shapes[0][1], shapes[0][2], C), x)
                    flat, off = cat.view(-1), 0
                    for q, s in zip(parts, shapes):
                        src, up = (q[1], 1) if isinstance(q, tuple) else (q, 0)
                        L.call('keep_slice_copy', src, flat[off:], s[0], s[1], s[2], s[3], s[3], C, up)
                        off += s[3]
                    y = cat
                elif kind == 'Detect':
                    feats = [ys[j] for j in f]
                    rows = [3 * q.shape[1] * q.shape[2] for q in feats]
                    pred = ops.empty((N, sum(rows), NO), x)
                    row0 = 0
                    for k, q in enumerate(feats):
                        raw = self._c(q, f'{p}.m.{k}', act=L.ACT_NONE)
                        L.call('keep_yolo_decode', raw, pred, N, q.shape[1], q.shape[2], STRIDES[k], self.w['anchor_grid'][k].contiguous(),
                               row0, sum(rows))
                        row0 += rows[k]
                    return pred
                ys.append(y)
        raise AssertionError('no Detect layer')


class EngineYoloModel:
    """Drop-in for ``YoloDetector.detector`` (a yolo.py Model): ``model(images)`` with images [N,3,H,W] float in [0, 1] on the
    device -> ``(pred [N, anchors, 16], None)`` -- face_detector.py:127-131 reads ``[0]``.  ``stride`` is what ``_preprocess`` asks
    for (``check_img_size(.., s=self.detector.stride.max())``, face_detector.py:58)."""

    def __init__(self, engine):
        self.engine = engine
        self.stride = torch.tensor(STRIDES)

    @classmethod
    def from_module(cls, module, device=None, precision='x3'):
        eps = {name: float(m.eps) for name, m in getattr(module, 'named_modules', lambda: [])() if isinstance(m, torch.nn.BatchNorm2d)}
        eng = YoloFaceEngine(module.state_dict(), precision=precision, bn_eps=eps)
        return cls(eng if device is None else eng.to(device))

    def __call__(self, images):
        if self.engine.w is None:
            self.engine.to(images.device)
        x = images.to(device=self.engine.device, dtype=torch.float32).permute(0, 2, 3, 1).contiguous()
        return self.engine.forward_nhwc(x), None

    def to(self, device):
        self.engine.to(device)
        return self

    def eval(self):
        return self

    def float(self):
        return self


def letterbox_geometry(h, w, stride=32):
    """Where ``_preprocess`` (face_detector.py:48-62, target_size None) puts a frame: ``check_img_size`` rounds the longer side up to a
    multiple of the largest stride, ``letterbox(auto=True, scaleup=True)`` (utils/datasets.py:5-33) scales the frame to it and pads both
    axes to the next multiple of 64 of the REMAINDER (np.mod(d, 64)), half on each side.  -> (rh, rw), (top, left), (H2, W2)."""
    size = int(math.ceil(max(h, w) / stride) * stride)
    r = min(size / h, size / w)
    rw, rh = int(round(w * r)), int(round(h * r))
    dw, dh = ((size - rw) % 64) / 2, ((size - rh) % 64) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return (rh, rw), (top, left), (rh + top + bottom, rw + left + right)


def faces_from_kept_rows(rows, net_hw, frame_hw, min_face):
    """What ``_postprocess`` + ``detect_faces`` make of one frame's suppressed detections (face_detector.py:80-104,133-141;
    scale_coords / scale_coords_landmarks / clip_coords, utils/general.py:42-63,249-272): rows float32 [k, >= 15]
    (x1 y1 x2 y2 conf lm x 10, network-input pixels, descending conf) -> int64 [k', 15] rows ``x1 y1 x2 y2 x1 lm x 10`` in frame pixels,
    or None.  torch CPU float32 arithmetic, as the reference's (its ``det = pred[image_id].cpu()``); the ``.round()`` of :85-86 is not in
    place and has no effect; ``int()`` truncates."""
    if rows is None or len(rows) == 0:
        return None
    det = torch.as_tensor(np.ascontiguousarray(rows[:, :15]), dtype=torch.float32).clone()
    (H2, W2), (H, W) = net_hw, frame_hw
    gain = min(H2 / H, W2 / W)
    pad_x, pad_y = (W2 - W * gain) / 2, (H2 - H * gain) / 2
    xs, ys = [0, 2, 5, 7, 9, 11, 13], [1, 3, 6, 8, 10, 12, 14]
    det[:, xs] = ((det[:, xs] - pad_x) / gain).clamp_(0, W)
    det[:, ys] = ((det[:, ys] - pad_y) / gain).clamp_(0, H)
    size = torch.tensor([W, H])
    box = (det[:, :4] / size.repeat(2)).double() * size.repeat(2).double()          # (/ gn in float32, * width in Python floats)
    lms = (det[:, 5:15] / size.repeat(5)).double() * size.repeat(5).double()
    box, lms = box.trunc().long(), lms.trunc().long()
    big = ~((box[:, 3] - box[:, 1]) < min_face)
    if not bool(big.any()):
        return None
    box, lms = box[big], lms[big]
    return torch.cat((box, box[:, :1], lms), dim=1).numpy()


def yolo_detect_batch_device(det, frames_bgr, conf_thres=0.7, iou_thres=0.5, max_frames=16, cap=1024):
    """``yolo_detect_batch`` with everything between the uint8 frames and the kept detections on the device.  frames: uint8 [N,H,W,3] BGR
    (numpy or tensor, host or device).  Per chunk of ``max_frames``: H2D of the uint8 frames, ``keep_yolo_letterbox_u8`` (BGR2RGB +
    resize + 114 border + / 255 -> NHWC float), the network, ``keep_yolo_select`` (objectness and conf thresholds, xywh2xyxy -> compact
    list), ``keep_retina_nms`` (= torchvision.ops.nms); one D2H of the counts, one of the kept rows.  A frame whose compact list overflowed
    (more than ``cap`` candidates: a threshold near 0) or that holds two candidates with equal conf is finished on the host from its device
    tensors with the same arithmetic (``engine/retinaface.py:nms``)."""
    from .retinaface import nms as host_nms
    eng = det.detector.engine
    frames = torch.as_tensor(np.ascontiguousarray(frames_bgr) if isinstance(frames_bgr, np.ndarray) else frames_bgr)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError(f'yolo_detect_batch_device: uint8 [N,H,W,3] frames expected, got {frames.dtype} {tuple(frames.shape)}')
    N, H, W, _ = frames.shape
    if eng.w is None:
        eng.to(frames.device if frames.device.type == 'cuda' else getattr(det, 'device', 'cuda'))
    dev = eng.device
    (rh, rw), (top, left), (H2, W2) = letterbox_geometry(H, W, int(max(STRIDES)))
    min_face = getattr(det, 'min_face', 10)
    out = []
    for s0 in range(0, N, max_frames):
        chunk = frames[s0:s0 + max_frames]
        if chunk.device.type != 'cuda':      # pageable -> device directly: 35 MB of frames in 0.63 ms (55 GB/s); a pinned staging copy in front of it
            chunk = chunk.to(dev)            # (torch's 128-thread copy_) stalls for 70-80 ms every few calls, profiles/r05_h2d_staging.txt
        chunk = chunk.contiguous()
        n = chunk.shape[0]
        with torch.cuda.device(dev):
            x = torch.empty((n, H2, W2, 3), dtype=torch.float32, device=dev)
            L.call('keep_yolo_letterbox_u8', chunk, x, n, H, W, rh, rw, top, left, H2, W2, 1)
            pred = eng.forward_nhwc(x)
            P = pred.shape[1]
            dets = torch.empty((n, cap, 16), dtype=torch.float32, device=dev)
            counts = torch.zeros((n,), dtype=torch.int32, device=dev)
            L.call('keep_yolo_select', pred, dets, counts, n, P, cap, float(conf_thres))
            kept = torch.empty((n, cap, 16), dtype=torch.float32, device=dev)
            kcnt = torch.empty((n,), dtype=torch.int32, device=dev)
            L.call('keep_retina_nms', dets, counts, kept, kcnt, n, cap, float(iou_thres))
            kc = kcnt.cpu().numpy()
            top_k = int(max(0, kc.max()))
            rows_all = kept[:, :top_k].cpu().numpy() if top_k else None
        for i in range(n):
            if kc[i] >= 0:
                rows = rows_all[i, :kc[i]] if kc[i] else None
            else:                   # -1: the list overflowed, -2: equal conf values -> this frame's candidates are ordered / suppressed on the host
                if kc[i] == -2:
                    cand = dets[i, :int(counts[i])].cpu().numpy()
                    cand = cand[np.argsort(cand[:, 15], kind='stable')]      # arrival order -> row order: equal conf values keep the prediction's order
                else:
                    full = pred[i].cpu().numpy()
                    full = full[full[:, 4] > np.float32(conf_thres)]
                    conf = full[:, 15] * full[:, 4]
                    hw_, hh_ = full[:, 2] / np.float32(2), full[:, 3] / np.float32(2)
                    cand = np.concatenate((np.stack((full[:, 0] - hw_, full[:, 1] - hh_, full[:, 0] + hw_, full[:, 1] + hh_, conf), 1),
                                           full[:, 5:15]), 1)[conf > np.float32(conf_thres)]
                rows = cand[host_nms(cand[:, :5], np.float32(iou_thres))] if len(cand) else None
            out.append(faces_from_kept_rows(rows, (H2, W2), (H, W), min_face))
    return out


def yolo_detect_batch(det, frames_bgr, conf_thres=0.7, iou_thres=0.5):
    """``YoloDetector.detect_faces`` (face_detector.py:113-141) for a stack of equally sized frames with ONE network call per chunk, per
    frame: ``[x1, y1, x2, y2, x1, lm x 10]`` rows, or None for a frame without faces (detect_faces called on that frame alone returns
    None).  ``KEEPFaceProcessor._detect_all`` finds it as ``det.detect_batch``.  uint8 frames with the engine's network behind the detector
    and ``target_size`` None (detection/__init__.py:42-49 never sets it) take ``yolo_detect_batch_device``; ``KEEP_AMD_YOLO_DEVICE=0``,
    other frame types or a ``target_size`` keep the detector's own ``_preprocess`` (letterbox, cv2) and ``_postprocess`` (NMS, rescaling,
    min_face filter) on the whole list around the one network call."""
    dtype = getattr(frames_bgr, 'dtype', None)
    if (os.environ.get('KEEP_AMD_YOLO_DEVICE', '1') != '0' and isinstance(getattr(det, 'detector', None), EngineYoloModel)
            and not getattr(det, 'target_size', None) and dtype in (np.uint8, torch.uint8) and len(frames_bgr)):
        return yolo_detect_batch_device(det, frames_bgr, conf_thres, iou_thres)
    import copy
    import cv2
    images = [cv2.cvtColor(np.ascontiguousarray(img), cv2.COLOR_BGR2RGB) for img in frames_bgr]
    origimgs = copy.deepcopy(images)
    x = det._preprocess(images)
    with torch.no_grad():
        pred = det.detector(x)[0]
    bboxes, points = det._postprocess(x, origimgs, pred, conf_thres, iou_thres)
    out = []
    for b, p in zip(bboxes, points):
        if len(p) == 0:
            out.append(None)
            continue
        b = np.array(b).reshape(-1, 4)
        out.append(np.concatenate((b, b[:, 0].reshape(-1, 1), np.array(p).reshape(-1, 10)), axis=1))
    return out


def synth_yolo_state_dict(name='YOLOv5n', seed=0):
    """Deterministic synthetic weights: He-like convolutions, BatchNorm gamma 1 +- 0.1 (0.6 on a Bottleneck's second convolution so a
    stack of 9 residual blocks stays O(1)), beta / mean +- 0.1, var in [0.7, 1.3]; Detect: small weights, objectness / class biases
    around 0 so that sigmoid spans its range, anchors = the yaml's."""
    from .synth import uniform_pm1
    out = {}
    for key, shape in yolo_state_dict_spec(name).items():
        leaf = key.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[key] = torch.tensor(100, dtype=torch.int64)
            continue
        if leaf == 'anchor_grid':
            out[key] = torch.tensor(ANCHORS, dtype=torch.float32).view(3, 1, 3, 1, 1, 2)
            continue
        if leaf == 'anchors':
            out[key] = torch.tensor(ANCHORS, dtype=torch.float32).view(3, 3, 2) / torch.tensor(STRIDES).view(3, 1, 1)
            continue
        n = int(np.prod(shape))
        u = uniform_pm1(f'yolo[{name}].' + key, n, seed)
        if len(shape) == 4:
            v = u * (math.sqrt(3.0) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3])))
            if '.m.' in key and key.split('.')[1] == str(len(YOLO_CFGS[name]) - 1):
                v = v * 6.0             # Detect: logits of a few units
        elif leaf == 'running_var':
            v = 1.0 + 0.3 * u
        elif leaf == 'weight':
            v = (0.6 if ('.m.' in key and '.cv2.bn.' in key) else 1.0) + 0.1 * u
        else:
            v = 0.1 * u
        out[key] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out
