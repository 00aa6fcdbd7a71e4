"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg are the only callers of anything under oracle/).

CPU restatements (plain PyTorch fp32) of the two face-analysis networks either side of the KEEP hot path:

  * ``parsenet_forward``    wm_facelib/parsing/parsenet.py:72-194 (ConvLayer / ResidualBlock / ParseNet.forward)
  * ``retinaface_forward``  wm_facelib/detection/retinaface/retinaface.py:83-146 (RetinaFace.forward, resnet50 and mobile0.25
                            configurations), retinaface_net.py:25-34,101-124 (conv_dw, MobileNetV1), 37-196 (SSH, FPN, heads) and
                            retinaface_utils.py (PriorBox, decode, NMS)

Pinned against the imported reference modules by ``oracle/make_golden_facelib.py`` -> ``tests/golden/facelib.npz``
(``tests/test_oracle_vs_golden.py``).  PARITY UNPINNED for the ResNet-50 trunk of RetinaFace: the reference takes it from
``torchvision.models.resnet50`` (retinaface.py:103-105), torchvision is not vendored under /root/reference and not installed
in the build image; the trunk below restates torchvision's published ResNet-50 v1.5 (Bottleneck with the stride on the 3x3
convolution, state-dict names ``body.conv1 / bn1 / layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1}``) and the golden
generator feeds the SAME restated trunk to the reference's FPN / SSH / heads.  The mobile0.25 configuration has no such gap:
MobileNetV1 is the reference's own class, and its golden is the output of the reference's modules from image to heads.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------------- ParseNet
def _conv_layer(x, W, p, scale='none', norm=False, act=False):
    """ConvLayer.forward, parsenet.py:101-109: [nearest x2] -> ReflectionPad2d(1) -> Conv2d -> [BatchNorm eval] -> [LeakyReLU 0.2]."""
    if scale == 'up':
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    x = F.pad(x, (1, 1, 1, 1), mode='reflect')
    x = F.conv2d(x, W[f'{p}.conv2d.weight'], W.get(f'{p}.conv2d.bias'), stride=2 if scale == 'down' else 1)
    if norm:
        q = f'{p}.norm.norm'
        x = F.batch_norm(x, W[f'{q}.running_mean'], W[f'{q}.running_var'], W[f'{q}.weight'], W[f'{q}.bias'], False, 0.0, BN_EPS)
    if act:
        x = F.leaky_relu(x, 0.2)
    return x


def _res_block(x, W, p, scale):
    """ResidualBlock.forward, parsenet.py:112-137."""
    conf = {'down': ('none', 'down'), 'up': ('up', 'none'), 'none': ('none', 'none')}[scale]
    idt = _conv_layer(x, W, f'{p}.shortcut_func', scale) if f'{p}.shortcut_func.conv2d.weight' in W else x
    h = _conv_layer(x, W, f'{p}.conv1', conf[0], norm=True, act=True)
    h = _conv_layer(h, W, f'{p}.conv2', conf[1], norm=True, act=False)
    return idt + h


def parsenet_forward(x, W, blocks):
    """ParseNet.forward (parsenet.py:188-194) -> out_mask [N,19,S,S].  ``blocks`` = engine.parsenet.parsenet_spec(...)."""
    feat = None
    for name, kind, _, _ in blocks:
        if name == 'out_mask_conv':
            return _conv_layer(x, W, name)
        if kind == 'conv':
            x = _conv_layer(x, W, name)
            continue
        if kind == 'none' and feat is None:
            feat = x
        x = _res_block(x, W, name, kind)
        if kind == 'none' and name == [b[0] for b in blocks if b[1] == 'none'][-1]:
            x = feat + x
    raise AssertionError('no out_mask_conv in the block list')


# ------------------------------------------------------------------------------------------------------- RetinaFace (resnet50, mobile0.25)
def _cbr(x, W, conv, bn, stride=1, pad=0, relu=True, leaky=0.0, groups=1):
    x = F.conv2d(x, W[f'{conv}.weight'], None, stride=stride, padding=pad, groups=groups)
    x = F.batch_norm(x, W[f'{bn}.running_mean'], W[f'{bn}.running_var'], W[f'{bn}.weight'], W[f'{bn}.bias'], False, 0.0, BN_EPS)
    if not relu:
        return x
    return F.leaky_relu(x, leaky) if leaky else F.relu(x)


MNET_STAGES = (('stage1', ((3, 8, 2), (8, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1))),
               ('stage2', ((64, 128, 2),) + ((128, 128, 1),) * 5),
               ('stage3', ((128, 256, 2), (256, 256, 1))))


def mobilenet_trunk(x, W):
    """MobileNetV1 x0.25 as RetinaFace('mobile0.25') uses it (retinaface_net.py:101-124 through IntermediateLayerGetter with
    return_layers stage1 / stage2 / stage3, retinaface.py:37-41,96-98): conv_bn(3, 8, stride 2, leaky 0.1), then conv_dw blocks =
    depthwise 3x3 (groups = channels) + BN + LeakyReLU(0.1), 1x1 + BN + LeakyReLU(0.1) (retinaface_net.py:25-34).
    Pinned: tests/golden/facelib.npz holds the reference modules' own outputs (oracle/make_golden_facelib.py)."""
    feats = []
    for stage, blocks in MNET_STAGES:
        for i, (ci, co, stride) in enumerate(blocks):
            p = f'body.{stage}.{i}'
            if ci == 3:
                x = _cbr(x, W, f'{p}.0', f'{p}.1', stride=stride, pad=1, leaky=0.1)
            else:
                x = _cbr(x, W, f'{p}.0', f'{p}.1', stride=stride, pad=1, leaky=0.1, groups=ci)
                x = _cbr(x, W, f'{p}.3', f'{p}.4', leaky=0.1)
        feats.append(x)
    return feats


def resnet50_trunk(x, W):
    """torchvision ResNet-50 v1.5 up to layer4 (IntermediateLayerGetter(return_layers layer2/3/4), retinaface.py:104-105):
    conv1 7x7 s2 p3 -> bn -> relu -> maxpool 3 s2 p1 -> Bottleneck stacks [3, 4, 6, 3], stride on the 3x3 convolution.
    PARITY UNPINNED (torchvision is not available; see the module docstring)."""
    x = _cbr(x, W, 'body.conv1', 'body.bn1', stride=2, pad=3)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for name, n, stride in (('layer1', 3, 1), ('layer2', 4, 2), ('layer3', 6, 2), ('layer4', 3, 2)):
        for i in range(n):
            p = f'body.{name}.{i}'
            s = stride if i == 0 else 1
            idt = _cbr(x, W, f'{p}.downsample.0', f'{p}.downsample.1', stride=s, relu=False) if i == 0 else x
            h = _cbr(x, W, f'{p}.conv1', f'{p}.bn1')
            h = _cbr(h, W, f'{p}.conv2', f'{p}.bn2', stride=s, pad=1)
            h = _cbr(h, W, f'{p}.conv3', f'{p}.bn3', relu=False)
            x = F.relu(h + idt)
        if name != 'layer1':
            feats.append(x)
    return feats


def retinaface_forward(x, W, backbone='resnet50'):
    """RetinaFace.forward, phase 'test' (retinaface.py:124-146) -> (bbox_regressions [N,P,4], softmax conf [N,P,2], landmarks [N,P,10])."""
    f = mobilenet_trunk(x, W) if backbone == 'mobile0.25' else resnet50_trunk(x, W)
    # FPN, retinaface_net.py:66-98 (out_channels 256 -> leaky 0 = ReLU; 64 -> LeakyReLU(0.1), lines 74-76; SSH likewise, 41-43)
    lk = 0.1 if W['fpn.output1.0.weight'].shape[0] <= 64 else 0.0
    o1 = _cbr(f[0], W, 'fpn.output1.0', 'fpn.output1.1', leaky=lk)
    o2 = _cbr(f[1], W, 'fpn.output2.0', 'fpn.output2.1', leaky=lk)
    o3 = _cbr(f[2], W, 'fpn.output3.0', 'fpn.output3.1', leaky=lk)
    o2 = _cbr(o2 + F.interpolate(o3, size=o2.shape[2:], mode='nearest'), W, 'fpn.merge2.0', 'fpn.merge2.1', pad=1, leaky=lk)
    o1 = _cbr(o1 + F.interpolate(o2, size=o1.shape[2:], mode='nearest'), W, 'fpn.merge1.0', 'fpn.merge1.1', pad=1, leaky=lk)
    feats = []
    for k, o in enumerate((o1, o2, o3)):          # SSH, retinaface_net.py:37-63
        s = f'ssh{k + 1}'
        c3 = _cbr(o, W, f'{s}.conv3X3.0', f'{s}.conv3X3.1', pad=1, relu=False)
        c51 = _cbr(o, W, f'{s}.conv5X5_1.0', f'{s}.conv5X5_1.1', pad=1, leaky=lk)
        c5 = _cbr(c51, W, f'{s}.conv5X5_2.0', f'{s}.conv5X5_2.1', pad=1, relu=False)
        c72 = _cbr(c51, W, f'{s}.conv7X7_2.0', f'{s}.conv7X7_2.1', pad=1, leaky=lk)
        c7 = _cbr(c72, W, f'{s}.conv7x7_3.0', f'{s}.conv7x7_3.1', pad=1, relu=False)
        feats.append(F.relu(torch.cat([c3, c5, c7], 1)))

    def head(name, k):
        outs = []
        for i, ft in enumerate(feats):
            o = F.conv2d(ft, W[f'{name}.{i}.conv1x1.weight'], W[f'{name}.{i}.conv1x1.bias'])
            outs.append(o.permute(0, 2, 3, 1).contiguous().view(o.shape[0], -1, k))
        return torch.cat(outs, 1)
    return head('BboxHead', 4), F.softmax(head('ClassHead', 2), dim=-1), head('LandmarkHead', 10)


# ------------------------------------------------------------------------------------------------------- YOLOv5-face (n, l)
def _yconv(x, W, p, k=1, s=1, act=True, eps=BN_EPS):
    """Conv.forward (yolov5face/models/common.py:32-45): Conv2d(k, s, k // 2, bias=False) -> BatchNorm2d -> SiLU."""
    x = F.conv2d(x, W[f'{p}.conv.weight'], None, stride=s, padding=k // 2)
    x = F.batch_norm(x, W[f'{p}.bn.running_mean'], W[f'{p}.bn.running_var'], W[f'{p}.bn.weight'], W[f'{p}.bn.bias'], False, 0.0, eps)
    return F.silu(x) if act else x


def _ybn(x, W, p, eps=BN_EPS):
    return F.batch_norm(x, W[f'{p}.running_mean'], W[f'{p}.running_var'], W[f'{p}.weight'], W[f'{p}.bias'], False, 0.0, eps)


def _yshuffle(x, W, p, stride):
    """ShuffleV2Block.forward + channel_shuffle (common.py:17-22,103-155)."""
    def branch2(t):
        t = F.silu(_ybn(F.conv2d(t, W[f'{p}.branch2.0.weight']), W, f'{p}.branch2.1'))
        t = _ybn(F.conv2d(t, W[f'{p}.branch2.3.weight'], None, stride, 1, groups=t.shape[1]), W, f'{p}.branch2.4')
        return F.silu(_ybn(F.conv2d(t, W[f'{p}.branch2.5.weight']), W, f'{p}.branch2.6'))
    if stride == 1:
        x1, x2 = x.chunk(2, dim=1)
        out = torch.cat((x1, branch2(x2)), 1)
    else:
        b1 = _ybn(F.conv2d(x, W[f'{p}.branch1.0.weight'], None, stride, 1, groups=x.shape[1]), W, f'{p}.branch1.1')
        b1 = F.silu(_ybn(F.conv2d(b1, W[f'{p}.branch1.2.weight']), W, f'{p}.branch1.3'))
        out = torch.cat((b1, branch2(x)), 1)
    n, c, h, w = out.shape
    return out.view(n, 2, c // 2, h, w).transpose(1, 2).contiguous().view(n, c, h, w)


def _yc3(x, W, p, n, shortcut):
    """C3.forward with n Bottleneck(c_, c_, shortcut, e=1.0) (common.py:56-66,86-100)."""
    y = _yconv(x, W, f'{p}.cv1')
    for k in range(n):
        h = _yconv(_yconv(y, W, f'{p}.m.{k}.cv1'), W, f'{p}.m.{k}.cv2', k=3)
        y = y + h if shortcut else h
    return _yconv(torch.cat((y, _yconv(x, W, f'{p}.cv2')), 1), W, f'{p}.cv3')


def yolo_forward(x, W, layers, anchors, strides):
    """Model.forward_once + Detect.forward, inference (yolov5face/models/yolo.py:44-84,133-142) on a restated layer list
    ``layers`` = engine/yoloface.py:yolo_layers(name) (parse_model of the yaml) -> pred [N, anchors, 16].
    Pinned: tests/golden/facelib.npz holds the outputs of the reference's own Model(yaml) (oracle/make_golden_facelib.py)."""
    ys = []
    y = x
    for i, f, kind, n, c1, c2, args in layers:
        p = f'model.{i}'
        if kind == 'StemBlock':                      # common.py:47-61
            s1 = _yconv(y, W, f'{p}.stem_1', k=3, s=2)
            b = _yconv(_yconv(s1, W, f'{p}.stem_2a'), W, f'{p}.stem_2b', k=3, s=2)
            y = _yconv(torch.cat((b, F.max_pool2d(s1, 2, 2, ceil_mode=True)), 1), W, f'{p}.stem_3')
        elif kind == 'Conv':
            y = _yconv(y, W, p, k=args[1], s=args[2])
        elif kind == 'C3':
            y = _yc3(y, W, p, n, args[1])
        elif kind == 'SPP':                          # common.py:157-166
            t = _yconv(y, W, f'{p}.cv1')
            y = _yconv(torch.cat([t] + [F.max_pool2d(t, k, 1, k // 2) for k in args[1]], 1), W, f'{p}.cv2')
        elif kind == 'Shuffle':
            for k in range(n):
                y = _yshuffle(y, W, f'{p}.{k}' if n > 1 else p, args[1])
        elif kind == 'Up':
            y = F.interpolate(y, scale_factor=2, mode='nearest')
        elif kind == 'Concat':
            y = torch.cat([y if j == -1 else ys[j] for j in f], 1)
        elif kind == 'Detect':
            z = []
            for k, j in enumerate(f):
                r = F.conv2d(ys[j], W[f'{p}.m.{k}.weight'], W[f'{p}.m.{k}.bias'])
                bs, _, ny, nx = r.shape
                r = r.view(bs, 3, 16, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
                yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing='ij')
                grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
                ag = torch.tensor(anchors[k], dtype=torch.float32).view(1, 3, 1, 1, 2)
                o = torch.zeros_like(r)
                o[..., [0, 1, 2, 3, 4, 15]] = r[..., [0, 1, 2, 3, 4, 15]].sigmoid()
                o[..., 0:2] = (o[..., 0:2] * 2.0 - 0.5 + grid) * strides[k]
                o[..., 2:4] = (o[..., 2:4] * 2) ** 2 * ag
                for q in range(5):
                    o[..., 5 + 2 * q:7 + 2 * q] = r[..., 5 + 2 * q:7 + 2 * q] * ag + grid * strides[k]
                z.append(o.view(bs, -1, 16))
            return torch.cat(z, 1)
        ys.append(y)
    raise AssertionError('no Detect layer')
