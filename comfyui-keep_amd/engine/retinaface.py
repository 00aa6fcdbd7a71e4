"""RetinaFace (resnet50 and mobile0.25 configurations) on the MI355X kernels, batched over frames -- SURVEY.md 8f-4.

Reference: ``wm_facelib/detection/retinaface/retinaface.py:83-146`` (RetinaFace.__init__ / forward), ``retinaface_net.py:37-196``
(SSH, FPN, ClassHead / BboxHead / LandmarkHead), ``retinaface_utils.py:8-36,254-295`` (PriorBox, decode, decode_landm) and the
per-frame driver ``detect_faces`` (retinaface.py:208-256) that ``FaceRestoreHelper.get_face_landmarks_5`` calls once per video
frame with a host round trip each (face_restoration_helper.py:221, keep_processor.py:207-213).  The ResNet-50 trunk comes from
torchvision in the reference (retinaface.py:103-105); torchvision is not part of /root/reference, so the trunk follows
torchvision's published ResNet-50 v1.5 (stride on the 3x3 convolution of a Bottleneck) and its state-dict names
(``body.conv1 / bn1 / layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1}``).

How it maps onto the engine
  * every Conv2d + BatchNorm2d(eval) pair is folded into one ``keep_conv2d`` (weights * gamma / sqrt(var + eps)) with the
    ReLU in its epilogue; a Bottleneck's ``relu(conv3 + identity)`` is the fused residual epilogue + ``keep_act_inplace``;
  * stem: 7x7 stride-2 convolution (flattened-K kernel, Cin = 3) + ``keep_maxpool3s2``;
  * FPN: 1x1 laterals, ``keep_upsample_add`` (nearest resize to the finer map + sum), 3x3 merges;
  * SSH: the three branches write their channel slices of ONE [N,H,W,256] buffer (``out_ld``), the trailing ReLU of
    ``relu(cat(..))`` is the epilogue of each branch's last convolution;
  * the nine 1x1 heads of a level (2 anchors x (2 class + 4 box + 10 landmark) = 32 channels) are ONE 256 -> 32 GEMM whose
    NHWC output is already the ``permute(0, 2, 3, 1)`` layout the reference reshapes to [N, anchors, k];
  * ``retinaface_mobile0.25`` (detection/__init__.py:38-41; MobileNetV1 is part of the reference tree, retinaface_net.py:101-134,
    so this configuration is pinned END TO END against the reference's own modules): conv_dw = ``keep_dwconv3x3`` (depthwise 3x3
    + folded BatchNorm + LeakyReLU(0.1)) followed by a 1x1 ``keep_conv2d`` with the LeakyReLU(0.1) in its epilogue; FPN / SSH take
    ``leaky = 0.1`` because out_channel = 64 (retinaface_net.py:41-43,74-76); 64 -> 32 fused heads;
  * frames ride the batch axis; logits come back in one D2H copy per chunk of frames and the (tiny) softmax / prior decoding /
    NMS run in numpy on the host, restating retinaface_utils.py (``torchvision.ops.nms`` = greedy IoU suppression).
"""
import math
import os
from itertools import product

import numpy as np
import torch

from . import hiplib as L
from . import ops
from .weights import pack_blob, views

BN_EPS = 1e-5
CFG_RE50 = {'min_sizes': [[16, 32], [64, 128], [256, 512]], 'steps': [8, 16, 32], 'variance': [0.1, 0.2], 'clip': False,
            'in_channel': 256, 'out_channel': 256}          # retinaface.py:49-71
CFG_MNET = {'min_sizes': [[16, 32], [64, 128], [256, 512]], 'steps': [8, 16, 32], 'variance': [0.1, 0.2], 'clip': False,
            'in_channel': 32, 'out_channel': 64}            # retinaface.py:22-44
MEAN_BGR = (104.0, 117.0, 123.0)                             # retinaface.py:96-97
LAYERS = (('layer1', 64, 3, 1), ('layer2', 128, 4, 2), ('layer3', 256, 6, 2), ('layer4', 512, 3, 2))   # ResNet-50
# MobileNetV1 x0.25 (retinaface_net.py:101-124): (cin, cout, stride); the first entry of stage1 is a plain conv_bn, the rest conv_dw
MNET_STAGES = (('stage1', ((3, 8, 2), (8, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1))),
               ('stage2', ((64, 128, 2),) + ((128, 128, 1),) * 5),
               ('stage3', ((128, 256, 2), (256, 256, 1))))
BACKBONES = {'resnet50': CFG_RE50, 'mobile0.25': CFG_MNET}


def backbone_of(state_dict):
    """Which of the two configurations a state dict belongs to (by the names of its trunk)."""
    keys = {k.replace('module.', '') for k in state_dict}
    if 'body.stage1.0.0.weight' in keys:
        return 'mobile0.25'
    if 'body.conv1.weight' in keys:
        return 'resnet50'
    raise RuntimeError("RetinaFaceEngine: neither a RetinaFace(resnet50) nor a RetinaFace(mobile0.25) state dict")


def retinaface_state_dict_spec(backbone='resnet50'):
    """name -> shape of RetinaFace(backbone).state_dict() (BatchNorm num_batches_tracked included; the classifier of MobileNetV1
    is dropped by IntermediateLayerGetter, retinaface.py:97-98)."""
    spec = {}

    def bn(p, c):
        for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
            spec[f'{p}.{leaf}'] = (c,)
        spec[f'{p}.num_batches_tracked'] = ()
    cfg = BACKBONES[backbone]
    if backbone == 'mobile0.25':
        for stage, blocks in MNET_STAGES:
            for i, (ci, co, _) in enumerate(blocks):
                p = f'body.{stage}.{i}'
                if ci == 3:                                   # conv_bn (retinaface_net.py:6-9)
                    spec[f'{p}.0.weight'] = (co, ci, 3, 3)
                    bn(f'{p}.1', co)
                else:                                         # conv_dw (retinaface_net.py:25-34)
                    spec[f'{p}.0.weight'] = (ci, 1, 3, 3)
                    bn(f'{p}.1', ci)
                    spec[f'{p}.3.weight'] = (co, ci, 1, 1)
                    bn(f'{p}.4', co)
    else:
        spec['body.conv1.weight'] = (64, 3, 7, 7)
        bn('body.bn1', 64)
        cin = 64
        for name, width, n, _ in LAYERS:
            for i in range(n):
                p = f'body.{name}.{i}'
                spec[f'{p}.conv1.weight'] = (width, cin, 1, 1)
                bn(f'{p}.bn1', width)
                spec[f'{p}.conv2.weight'] = (width, width, 3, 3)
                bn(f'{p}.bn2', width)
                spec[f'{p}.conv3.weight'] = (4 * width, width, 1, 1)
                bn(f'{p}.bn3', 4 * width)
                if i == 0:
                    spec[f'{p}.downsample.0.weight'] = (4 * width, cin, 1, 1)
                    bn(f'{p}.downsample.1', 4 * width)
                cin = 4 * width
    oc = cfg['out_channel']
    for k, c in enumerate((cfg['in_channel'] * 2, cfg['in_channel'] * 4, cfg['in_channel'] * 8)):
        spec[f'fpn.output{k + 1}.0.weight'] = (oc, c, 1, 1)
        bn(f'fpn.output{k + 1}.1', oc)
    for m in ('merge1', 'merge2'):
        spec[f'fpn.{m}.0.weight'] = (oc, oc, 3, 3)
        bn(f'fpn.{m}.1', oc)
    for s in ('ssh1', 'ssh2', 'ssh3'):
        for nm, ci, co in (('conv3X3', oc, oc // 2), ('conv5X5_1', oc, oc // 4), ('conv5X5_2', oc // 4, oc // 4),
                           ('conv7X7_2', oc // 4, oc // 4), ('conv7x7_3', oc // 4, oc // 4)):
            spec[f'{s}.{nm}.0.weight'] = (co, ci, 3, 3)
            bn(f'{s}.{nm}.1', co)
    for head, k in (('ClassHead', 2), ('BboxHead', 4), ('LandmarkHead', 10)):
        for i in range(3):
            spec[f'{head}.{i}.conv1x1.weight'] = (2 * k, oc, 1, 1)
            spec[f'{head}.{i}.conv1x1.bias'] = (2 * k,)
    return spec


def _fold(sd, conv, bn):
    w = sd[f'{conv}.weight'].double()
    g, b = sd[f'{bn}.weight'].double(), sd[f'{bn}.bias'].double()
    m, v = sd[f'{bn}.running_mean'].double(), sd[f'{bn}.running_var'].double()
    s = g / torch.sqrt(v + BN_EPS)
    return (w * s.view(-1, 1, 1, 1)).float(), (b - m * s).float()


def prior_boxes(h, w, cfg=CFG_RE50):      # (both configurations share min_sizes / steps / clip)
    """PriorBox(cfg, image_size=(h, w)).forward() (retinaface_utils.py:8-36) as a float32 [P,4] array (cx, cy, w, h)."""
    anchors = []
    for k, step in enumerate(cfg['steps']):
        fh, fw = math.ceil(h / step), math.ceil(w / step)
        for i, j in product(range(fh), range(fw)):
            for ms in cfg['min_sizes'][k]:
                anchors += [(j + 0.5) * step / w, (i + 0.5) * step / h, ms / w, ms / h]
    out = np.asarray(anchors, dtype=np.float32).reshape(-1, 4)
    return np.clip(out, 0, 1) if cfg['clip'] else out


def decode_boxes(loc, priors, variances):
    """retinaface_utils.py:254-271 on float32 numpy arrays."""
    boxes = np.concatenate((priors[:, :2] + loc[:, :2] * np.float32(variances[0]) * priors[:, 2:],
                            priors[:, 2:] * np.exp(loc[:, 2:] * np.float32(variances[1]))), 1).astype(np.float32)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def decode_landmarks(pre, priors, variances):
    """retinaface_utils.py:274-294."""
    return np.concatenate([priors[:, :2] + pre[:, 2 * k:2 * k + 2] * np.float32(variances[0]) * priors[:, 2:] for k in range(5)],
                          1).astype(np.float32)


def nms(dets, thresh):
    """torchvision.ops.nms as py_cpu_nms uses it (retinaface_utils.py:39-47): boxes sorted by score, a box is dropped when its
    IoU with a kept, higher-scoring box exceeds ``thresh``.  Returns kept indices in descending-score order.
    Up to 2048 candidates the pairwise IoUs are computed once (the same float32 element-wise arithmetic as the row-by-row form
    below, hence the same decisions) and the greedy pass is a loop over a boolean matrix: ~0.1 ms instead of ~1 ms per frame for the
    ~120 candidates a frame of the synthetic-weight detector yields."""
    x1, y1, x2, y2, sc = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-sc, kind='stable')
    n = len(order)
    if 1 < n <= 2048:
        X1, Y1, X2, Y2, A = x1[order], y1[order], x2[order], y2[order], areas[order]
        inter = (np.clip(np.minimum(X2[:, None], X2[None, :]) - np.maximum(X1[:, None], X1[None, :]), 0, None)
                 * np.clip(np.minimum(Y2[:, None], Y2[None, :]) - np.maximum(Y1[:, None], Y1[None, :]), 0, None))
        over = (inter / (A[:, None] + A[None, :] - inter)) > thresh          # over[i, j]: j is suppressed when i is kept
        alive = np.ones(n, bool)
        keep = []
        for i in range(n):
            if alive[i]:
                keep.append(int(order[i]))
                alive[i + 1:] &= ~over[i, i + 1:]
        return keep
    keep = []
    while order.size:
        i = order[0]
        keep.append(int(i))
        xx1, yy1 = np.maximum(x1[i], x1[order[1:]]), np.maximum(y1[i], y1[order[1:]])
        xx2, yy2 = np.minimum(x2[i], x2[order[1:]]), np.minimum(y2[i], y2[order[1:]])
        inter = np.clip(xx2 - xx1, 0, None) * np.clip(yy2 - yy1, 0, None)
        iou = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[1:][iou <= thresh]
    return keep


class RetinaFaceEngine:
    def __init__(self, state_dict, precision='x3', backbone=None):
        self.backbone = backbone or backbone_of(state_dict)
        self.cfg = BACKBONES[self.backbone]
        # LeakyReLU(0.1) in the FPN / SSH when out_channel <= 64 (retinaface_net.py:41-43,74-76), ReLU (= LeakyReLU(0)) otherwise
        self._act = L.ACT_LRELU01 if self.cfg['out_channel'] <= 64 else L.ACT_RELU
        spec = retinaface_state_dict_spec(self.backbone)
        sd = {k.replace('module.', ''): v for k, v in state_dict.items()}
        missing = [k for k in spec if k not in sd]
        bad = [k for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
        if missing or bad:
            raise RuntimeError(f"RetinaFaceEngine: not a RetinaFace({self.backbone}) state dict: missing {missing[:4]}, shapes {bad[:4]}")
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        t = {}

        def put(name, w, b):
            w = w.permute(0, 2, 3, 1).contiguous()
            if w.shape[1] == 1 and w.shape[2] == 1:
                w = w.reshape(w.shape[0], w.shape[3])
            t[f'{name}.weight'], t[f'{name}.bias'] = w, b.contiguous()
        if self.backbone == 'mobile0.25':
            for stage, blocks in MNET_STAGES:
                for i, (ci, _, _) in enumerate(blocks):
                    p = f'body.{stage}.{i}'
                    if ci == 3:
                        put(f'{stage}.{i}', *_fold(sd, f'{p}.0', f'{p}.1'))
                    else:
                        w, b = _fold(sd, f'{p}.0', f'{p}.1')                      # depthwise [C,1,3,3] -> tap-major [3,3,C]
                        t[f'{stage}.{i}.dw.weight'], t[f'{stage}.{i}.dw.bias'] = w[:, 0].permute(1, 2, 0).contiguous(), b.contiguous()
                        put(f'{stage}.{i}.pw', *_fold(sd, f'{p}.3', f'{p}.4'))
        else:
            put('stem', *_fold(sd, 'body.conv1', 'body.bn1'))
            for name, width, n, _ in LAYERS:
                for i in range(n):
                    p = f'body.{name}.{i}'
                    for c in (1, 2, 3):
                        put(f'{name}.{i}.conv{c}', *_fold(sd, f'{p}.conv{c}', f'{p}.bn{c}'))
                    if i == 0:
                        put(f'{name}.{i}.down', *_fold(sd, f'{p}.downsample.0', f'{p}.downsample.1'))
        for k in (1, 2, 3):
            put(f'fpn.output{k}', *_fold(sd, f'fpn.output{k}.0', f'fpn.output{k}.1'))
        for m in ('merge1', 'merge2'):
            put(f'fpn.{m}', *_fold(sd, f'fpn.{m}.0', f'fpn.{m}.1'))
        for s in ('ssh1', 'ssh2', 'ssh3'):
            for nm in ('conv3X3', 'conv5X5_1', 'conv5X5_2', 'conv7X7_2', 'conv7x7_3'):
                put(f'{s}.{nm}', *_fold(sd, f'{s}.{nm}.0', f'{s}.{nm}.1'))
        for i in range(3):      # one GEMM per level: channels [class 4 | box 8 | landmark 20]
            w = torch.cat([sd[f'{h}.{i}.conv1x1.weight'] for h in ('ClassHead', 'BboxHead', 'LandmarkHead')], 0)
            b = torch.cat([sd[f'{h}.{i}.conv1x1.bias'] for h in ('ClassHead', 'BboxHead', 'LandmarkHead')], 0)
            put(f'heads.{i}', w, b)
        self._blob, self._index = pack_blob(t)
        self.precision = precision
        self.device = torch.device('cpu')
        self.w = None
        self.o = ops.Ops()
        self._priors = {}
        self._priors_dev = {}
        self.max_survivors = int(os.environ.get('KEEP_AMD_DETECT_SURVIVORS', '4096'))      # rows of the device-side compact list per frame
        self.max_frames = int(os.environ.get('KEEP_AMD_DETECT_BATCH', '32'))
        # score ordering + NMS of the survivors on the device (keep_retina_nms); 0: the numpy path of round 3 (the tests' reference)
        self.device_nms = os.environ.get('KEEP_AMD_DEVICE_NMS', '1') != '0' and self.max_survivors <= 4096

    def to(self, device):
        device = torch.device(device)
        self._priors_dev = {}
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
        self._mean = torch.tensor(MEAN_BGR, dtype=torch.float32, device=device)
        if self.precision == 'x3':
            names = [n for n, (_, sh) in self._index.items() if len(sh) >= 2 and sh[-1] % 16 == 0 and not n.endswith('.dw.weight')]
            bx, table = ops.make_x3_blob(self._dev, self._index, self.w, names)       # one power-of-two scale per tensor
            self.o.set_precision(L.MMA_X3, self._dev, None, bx, 1.0, x3_scales=table)
        else:
            self.o.set_precision(L.MMA_F32, self._dev, None)
        return self

    # ------------------------------------------------------------------ network
    def _c(self, x, name, **kw):
        w = self.w[f'{name}.weight']
        if w.dim() == 2:
            stride = kw.pop('stride', 1)
            if stride == 1:
                return self.o.linear(x, w, self.w[f'{name}.bias'], **kw)
            return self.o.conv(x, w.view(w.shape[0], 1, 1, w.shape[1]), self.w[f'{name}.bias'], stride=stride, pad=0, ksize=1, **kw)
        return self.o.conv(x, w, self.w[f'{name}.bias'], ksize=w.shape[1], **kw)

    def _relu_(self, x):
        L.call('keep_act_inplace', x, x.numel(), L.ACT_RELU)
        return x

    @torch.no_grad()
    def forward_nhwc(self, x):
        """x [N,H,W,3] fp32 NHWC, BGR minus the channel means -> list of 3 level outputs [N,h,w,32] (class 4 | box 8 | landmark 20
        per pixel, two anchors each), finest level first."""
        if self.w is None:
            raise RuntimeError("RetinaFaceEngine: call .to('cuda') first")
        with torch.cuda.device(self.device):
            self.o.begin_forward(self.device)
            act = self._act
            feats = self._mobilenet(x.contiguous()) if self.backbone == 'mobile0.25' else self._resnet50(x.contiguous())
            # FPN (retinaface_net.py:79-98)
            o1, o2, o3 = (self._c(f, f'fpn.output{k + 1}', act=act) for k, f in enumerate(feats))

            def up_add(a, b):
                out = torch.empty_like(a)
                L.call('keep_upsample_add', a, b, out, a.shape[0], a.shape[1], a.shape[2], b.shape[1], b.shape[2], a.shape[3])
                return out
            o2 = self._c(up_add(o2, o3), 'fpn.merge2', pad=1, act=act)
            o1 = self._c(up_add(o1, o2), 'fpn.merge1', pad=1, act=act)
            outs = []
            for k, f in enumerate((o1, o2, o3)):
                s = f'ssh{k + 1}'
                N, H, W, C = f.shape
                cat = ops.empty((N, H, W, C), f)
                flat = cat.view(-1)
                # relu(cat[conv3X3 | conv5X5 | conv7X7]) == cat of the three ReLUs: each branch's last conv writes its slice
                self._c(f, f'{s}.conv3X3', pad=1, act=L.ACT_RELU, out=flat, out_ld=C)
                c5 = self._c(f, f'{s}.conv5X5_1', pad=1, act=act)
                self._c(c5, f'{s}.conv5X5_2', pad=1, act=L.ACT_RELU, out=flat[C // 2:], out_ld=C)
                c7 = self._c(c5, f'{s}.conv7X7_2', pad=1, act=act)
                self._c(c7, f'{s}.conv7x7_3', pad=1, act=L.ACT_RELU, out=flat[C // 2 + C // 4:], out_ld=C)
                outs.append(self._c(cat, f'heads.{k}'))
            return outs

    def _resnet50(self, x):
        """torchvision ResNet-50 v1.5 up to layer4 -> the layer2 / layer3 / layer4 maps (retinaface.py:100-102)."""
        y = self._c(x, 'stem', stride=2, pad=3, act=L.ACT_RELU)
        N, H, W, C = y.shape
        p = ops.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), y)
        L.call('keep_maxpool3s2', y, p, N, H, W, C)
        x = p
        feats = []
        for name, width, n, stride in LAYERS:
            for i in range(n):
                s = stride if i == 0 else 1
                idt = self._c(x, f'{name}.{i}.down', stride=s) if i == 0 else x
                h = self._c(x, f'{name}.{i}.conv1', act=L.ACT_RELU)
                h = self._c(h, f'{name}.{i}.conv2', stride=s, pad=1, act=L.ACT_RELU)
                x = self._relu_(self._c(h, f'{name}.{i}.conv3', residual=idt))
            if name != 'layer1':
                feats.append(x)
        return feats

    def _mobilenet(self, x):
        """MobileNetV1 x0.25 (retinaface_net.py:101-124) -> the stage1 / stage2 / stage3 maps (64, 128, 256 channels at strides
        8, 16, 32; return_layers of cfg_mnet, retinaface.py:37-41)."""
        feats = []
        for stage, blocks in MNET_STAGES:
            for i, (ci, co, stride) in enumerate(blocks):
                if ci == 3:
                    x = self._c(x, f'{stage}.{i}', stride=stride, pad=1, act=L.ACT_LRELU01)
                    continue
                N, H, W, C = x.shape
                d = ops.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, C), x)
                L.call('keep_dwconv3x3', x, self.w[f'{stage}.{i}.dw.weight'], self.w[f'{stage}.{i}.dw.bias'], d, N, H, W, C, stride,
                       L.ACT_LRELU01)
                x = self._c(d, f'{stage}.{i}.pw', act=L.ACT_LRELU01)
            feats.append(x)
        return feats

    def raw_heads(self, x_nhwc):
        """-> [N, P, 32]: the fused head convolutions of the three FPN levels, rows = pixels in the reference's prior order
        (level, y, x), per pixel [cls a0 a1 | box a0 a1 | landmarks a0 a1]."""
        outs = self.forward_nhwc(x_nhwc)
        N = outs[0].shape[0]
        return torch.cat([o.reshape(N, -1, 32) for o in outs], 1)

    def raw_outputs(self, x_nhwc):
        """-> (loc [N,P,4], class logits [N,P,2], landmarks [N,P,10]) in the reference's prior order (level, y, x, anchor)."""
        flat = self.raw_heads(x_nhwc)
        N = flat.shape[0]
        cls = flat[..., 0:4].reshape(N, -1, 2)
        loc = flat[..., 4:12].reshape(N, -1, 4)
        lm = flat[..., 12:32].reshape(N, -1, 10)
        return loc, cls, lm

    # ------------------------------------------------------------------ detection (retinaface.py:208-256, per frame of a batch)
    def detect_batch(self, frames_bgr_u8, conf_threshold=0.8, nms_threshold=0.4):
        """frames: uint8 (or float: converted to float32 like the reference does) [N,H,W,3] BGR (numpy or tensor, one size) -> list of float32 [n_i, 15] arrays (x1,y1,x2,y2,score,
        5 landmarks x,y): what ``detect_faces`` returns for each frame with ``use_origin_size=True``."""
        frames = torch.as_tensor(np.ascontiguousarray(frames_bgr_u8) if isinstance(frames_bgr_u8, np.ndarray) else frames_bgr_u8)
        N, H, W, _ = frames.shape
        key = (H, W)
        if key not in self._priors:
            self._priors[key] = prior_boxes(H, W, self.cfg)
        priors = self._priors[key]
        scale = np.array([W, H, W, H], np.float32)
        scale1 = np.array([W, H] * 5, np.float32)
        results = []
        cap = self.max_survivors
        for s in range(0, N, self.max_frames):
            chunk = frames[s:s + self.max_frames]
            if chunk.device.type != 'cuda':        # uint8 frames (4x fewer bytes than float frames), pageable -> device directly: 35 MB in 0.63 ms;
                chunk = chunk.to(self.device)      # the pinned staging copy of round 4 stalled for 70-80 ms every few calls (profiles/r05_h2d_staging.txt)
            chunk = chunk.contiguous()
            n = chunk.shape[0]
            with torch.cuda.device(self.device):
                if chunk.dtype == torch.uint8:
                    x = torch.empty((n, H, W, 3), dtype=torch.float32, device=self.device)
                    L.call('keep_u8_to_f32', chunk, x, chunk.numel())
                else:                                                             # 16-bit sources: read_image made them float64
                    x = chunk.to(torch.float32)                                   # (np.float32(image), retinaface.py:213)
                x = ops.add_bcast(x, self._mean, alpha=-1.0)                      # image - mean_tensor (retinaface.py:224)
                heads = self.raw_heads(x)                                         # [n, P, 32] on the device
                P = heads.shape[1]
                if key not in self._priors_dev:
                    self._priors_dev[key] = torch.from_numpy(priors).to(self.device)
                # scores, threshold, box / landmark decode on the device: only the survivors cross PCIe (retinaface.py:231-246)
                dets = torch.empty((n, cap, 16), dtype=torch.float32, device=self.device)
                counts = torch.zeros(n, dtype=torch.int32, device=self.device)
                L.call('keep_retina_decode', heads, self._priors_dev[key], dets, counts, n, P, cap,
                       float(self.cfg['variance'][0]), float(self.cfg['variance'][1]), float(W), float(H), float(conf_threshold))
                if self.device_nms:      # ordering + greedy IoU suppression on the device too: only the kept rows cross PCIe
                    kept = torch.empty_like(dets)
                    kcnt = torch.empty(n, dtype=torch.int32, device=self.device)
                    L.call('keep_retina_nms', dets, counts, kept, kcnt, n, cap, float(nms_threshold))
                    kc = kcnt.cpu().numpy()
                    kmax = int(kc.max(initial=0))
                    rows = kept[:, :kmax, :15].cpu().numpy() if kmax > 0 else np.zeros((n, 0, 15), np.float32)
                    over = [i for i in range(n) if kc[i] == -1]
                    if over:
                        flat_over = heads[over].float().cpu().numpy()
                    tied = [i for i in range(n) if kc[i] == -2]
                    if tied:             # two survivors with the same score bits: numpy's own (unstable) argsort decides their order
                        tied_rows = self._tied_frames(dets, counts, tied, cap, nms_threshold)
                    for i in range(n):
                        if kc[i] == -1:      # more survivors than the compact list holds: that frame's heads go to the host decoder
                            results.append(self._host_decode(flat_over[over.index(i)], priors, scale, scale1, conf_threshold, nms_threshold))
                        elif kc[i] == -2:
                            results.append(tied_rows[tied.index(i)])
                        else:
                            results.append(np.ascontiguousarray(rows[i, :kc[i]]))
                    continue
                cnt = counts.cpu().numpy()
                kmax = int(min(cnt.max(initial=0), cap))
                rows = dets[:, :kmax].cpu().numpy() if kmax else np.zeros((n, 0, 16), np.float32)
                over = [i for i in range(n) if cnt[i] > cap]
                if over:            # more survivors than the compact list holds (a threshold near 0): that frame's heads go to the host decoder
                    flat_over = heads[over].float().cpu().numpy()
            for i in range(n):
                if cnt[i] > cap:
                    results.append(self._host_decode(flat_over[over.index(i)], priors, scale, scale1, conf_threshold, nms_threshold))
                    continue
                results.append(self._host_order_nms(rows[i, :cnt[i]], nms_threshold))
        return results

    def _tied_frames(self, dets, counts, tied, cap, nms_threshold):
        """Frames with equal score bit patterns (``keep_retina_nms`` -> -2): the ORDER is made on the host exactly as the reference makes it
        (anchor order, then ``scores.argsort()[::-1]``: numpy's introsort decides the ties, retinaface.py:240) from the scores and anchor
        indices alone; the permutation goes up and the suppression + compaction run on the device (``keep_retina_nms_ordered``: the
        arithmetic of ``nms`` below).  One D2H of two columns, one H2D of the permutation, one launch, one D2H of the kept rows for all tied
        frames of the chunk -- instead of a round trip and an O(n^2) numpy pass per frame.  ``KEEP_AMD_DEVICE_TIED_NMS=0``: the numpy path."""
        idx = torch.tensor(tied, dtype=torch.int64, device=dets.device)
        cnt_t = counts[idx].contiguous()
        cnt = cnt_t.cpu().numpy()
        if os.environ.get('KEEP_AMD_DEVICE_TIED_NMS', '1') == '0':
            return [self._host_order_nms(dets[i, :c].cpu().numpy(), nms_threshold) for i, c in zip(tied, cnt)]
        cmax = int(cnt.max())
        d_t = dets[idx].contiguous()
        cols = d_t[:, :cmax, [4, 15]].cpu().numpy()                                   # scores, anchor indices
        order = np.zeros((len(tied), cap), np.int32)
        for k, c in enumerate(cnt):
            by_anchor = np.argsort(cols[k, :c, 1], kind='stable')                      # what np.where(scores > thr) yields
            order[k, :c] = by_anchor[cols[k, :c, 0][by_anchor].argsort()[::-1]]        # scores.argsort()[::-1] on that list
        kept = torch.empty_like(d_t)
        kcnt = torch.empty(len(tied), dtype=torch.int32, device=dets.device)
        L.call('keep_retina_nms_ordered', d_t, cnt_t, torch.from_numpy(order).to(dets.device), kept, kcnt, len(tied), cap, float(nms_threshold))
        kc = kcnt.cpu().numpy()
        rows = kept[:, :int(kc.max(initial=0)), :15].cpu().numpy()
        return [np.ascontiguousarray(rows[k, :kc[k]]) for k in range(len(tied))]

    @staticmethod
    def _host_order_nms(r, nms_threshold):
        """Device-decoded survivors of one frame [k, 16] (x1 y1 x2 y2 score lm x 10 anchor) -> ordered + suppressed [k', 15] with numpy,
        exactly as the reference's host code does (retinaface.py:240-246)."""
        r = r[np.argsort(r[:, 15], kind='stable')]                                 # anchor order: what np.where(scores > thr) yields
        order = r[:, 4].argsort()[::-1]                                            # retinaface.py:240: scores.argsort()[::-1]
        r = r[order]
        dets_i = np.ascontiguousarray(r[:, :5])
        keep = nms(dets_i, nms_threshold) if len(dets_i) else []
        return np.concatenate((dets_i[keep, :], r[keep, 5:15]), axis=1) if len(dets_i) else np.zeros((0, 15), np.float32)

    def _host_decode(self, flat, priors, scale, scale1, conf_threshold, nms_threshold):
        """One frame's head rows [P, 32] through the numpy decoder (the path of rounds 2-3; kept for frames with more survivors
        than ``max_survivors`` and as the reference of tests/test_gpu_facelib.py)."""
        cls = flat[:, 0:4].reshape(-1, 2)
        loc = flat[:, 4:12].reshape(-1, 4)
        lm = flat[:, 12:32].reshape(-1, 10)
        e = np.exp(cls - cls.max(1, keepdims=True))
        scores = (e[:, 1] / e.sum(1)).astype(np.float32)                          # F.softmax(classifications, -1)[:, 1]
        boxes = decode_boxes(loc, priors, self.cfg['variance']) * scale
        lms = decode_landmarks(lm, priors, self.cfg['variance']) * scale1
        inds = np.where(scores > conf_threshold)[0]
        boxes, lms, sc = boxes[inds], lms[inds], scores[inds]
        order = sc.argsort()[::-1]
        boxes, lms, sc = boxes[order], lms[order], sc[order]
        dets = np.hstack((boxes, sc[:, None])).astype(np.float32, copy=False)
        keep = nms(dets, nms_threshold) if len(dets) else []
        return np.concatenate((dets[keep, :], lms[keep]), axis=1) if len(dets) else np.zeros((0, 15), np.float32)


class EngineRetinaFace:
    """Drop-in for the helper's ``face_detector`` (an ``init_detection_model('retinaface_resnet50' | 'retinaface_mobile0.25')``
    RetinaFace, detection/__init__.py:34-41): the calls
    FaceRestoreHelper makes -- ``detect_faces(image, conf_threshold)`` -- plus ``detect_batch`` for the processor's batched
    pre-pass over a whole video (one engine call per chunk of frames instead of one detector call per frame)."""
    def __init__(self, engine):
        self.engine = engine
        self.backbone = {'resnet50': 'Resnet50', 'mobile0.25': 'mobilenet0.25'}[engine.backbone]      # cfg['name'], retinaface.py:24,47

    @classmethod
    def from_module(cls, module, device=None, precision='x3'):
        eng = RetinaFaceEngine(module.state_dict(), precision=precision)
        return cls(eng if device is None else eng.to(device))

    def detect_faces(self, image, conf_threshold=0.8, nms_threshold=0.4, use_origin_size=True):
        if not use_origin_size:
            raise NotImplementedError("EngineRetinaFace: the helper always passes use_origin_size=True (it resizes itself)")
        img = np.asarray(image)
        return self.engine.detect_batch(img[None], conf_threshold, nms_threshold)[0]

    def detect_batch(self, frames, conf_threshold=0.8, nms_threshold=0.4):
        return self.engine.detect_batch(frames, conf_threshold, nms_threshold)

    def to(self, device):
        self.engine.to(device)
        return self

    def eval(self):
        return self


def synth_retinaface_state_dict(seed=0, backbone='resnet50'):
    """Deterministic synthetic RetinaFace(resnet50 | mobile0.25) weights: He-like conv weights, BatchNorm gamma 1 +- 0.1 (0.5 on the last BN of
    a Bottleneck so the residual stream stays O(1) over 16 blocks), beta / mean +- 0.1, var in [0.7, 1.3]."""
    from .synth import uniform_pm1
    out = {}
    tag = 'retinaface.' if backbone == 'resnet50' else f'retinaface[{backbone}].'
    for name, shape in retinaface_state_dict_spec(backbone).items():
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[name] = torch.tensor(100, dtype=torch.int64)
            continue
        n = int(np.prod(shape))
        u = uniform_pm1(tag + name, n, seed)
        if len(shape) == 4:
            v = u * (math.sqrt(3.0) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3])))
            if name in ('body.conv1.weight', 'body.stage1.0.0.weight'):
                v = v * 0.01            # inputs are BGR - mean, O(100): bring the stem's output to O(1) like a trained BatchNorm does
            elif 'Head' in name:
                v = v * (0.2 if backbone == 'resnet50' else 2.0)   # logits / regressions of a few units: an informative softmax, exp() in range
        elif leaf == 'running_var':
            v = 1.0 + 0.3 * u
        elif leaf == 'weight':
            v = (0.5 if '.bn3.' in name else 1.0) + 0.1 * u
        else:
            v = 0.1 * u
        out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out
