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

  * ``yolo_preprocess`` / ``yolo_postprocess``  YoloDetector._preprocess / _postprocess / detect_faces (yolov5face/face_detector.py:48-141,
                            utils/datasets.py:5-36 letterbox, utils/general.py:42-165,249-272 scale_coords / non_max_suppression_face)

``tests/golden/yolo_prepost.npz`` holds what the reference's OWN ``YoloDetector.detect_faces`` returns (``make_golden_facelib.py
--yolo-prepost``) with its two absent dependencies replaced: PARITY UNPINNED for ``cv2.resize(INTER_LINEAR)`` on uint8 (OpenCV is not
installed; ``cv2_resize_linear_u8`` restates the fixed-point bilinear of OpenCV's modules/imgproc/src/resize.cpp, and the golden run uses
that same restatement as its ``cv2.resize``) and for ``torchvision.ops.nms`` (greedy IoU suppression, as for RetinaFace).  Everything else
on that path -- the letterbox geometry, the padding, the thresholds, xywh2xyxy, the rescaling to the frame, clamps, the integer
truncation, the min_face filter, the assembly of the result -- is the reference's own code in the golden run; the frames whose longer side
is a multiple of 32 are never resized, so for them the pre-processing golden has no unpinned part.
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


# ------------------------------------------------------------------------------------- YoloDetector pre / post-processing
def cv2_resize_linear_u8(img, dw, dh):
    """cv2.resize(img uint8 [H,W,C], (dw, dh), interpolation=cv2.INTER_LINEAR) -- OpenCV modules/imgproc/src/resize.cpp: resizeGeneric_
    with HResizeLinear<uchar,int,short> / VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>> (11-bit coefficients).  PARITY
    UNPINNED (OpenCV absent from the image).  An exact 2x reduction is INTER_AREA inside cv2.resize and is not restated."""
    H, W = img.shape[:2]
    if (dw, dh) == (W, H):
        return img.copy()
    if H == 2 * dh and W == 2 * dw:
        raise NotImplementedError('cv2.resize turns an exact 2x INTER_LINEAR reduction into INTER_AREA')

    def coef(dsize, ssize, clamp):
        scale = 1.0 / (dsize / ssize)                                    # double inv_scale = (double)dsize / ssize; scale = 1. / inv_scale
        f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = f - i.astype(np.float32)
        if clamp:                                                         # columns: the position is clamped and the weight reset
            lo, hi = i < 0, i >= ssize - 1
            i = np.where(lo, 0, np.where(hi, ssize - 1, i))
            f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        c0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
        c1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
        return i, c0, c1

    sx, a0, a1 = coef(dw, W, True)
    sy, b0, b1 = coef(dh, H, False)
    src = img.astype(np.int64)
    x1 = np.minimum(sx + 1, W - 1)
    hor = src[:, sx] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # [H, dw, C]
    r0, r1 = hor[np.clip(sy, 0, H - 1)], hor[np.clip(sy + 1, 0, H - 1)]              # rows are clamped, their weights kept
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def yolo_letterbox_geometry(h, w, stride=32):
    """check_img_size + letterbox(auto=True, scaleup=True) (face_detector.py:58-59 with target_size None, utils/datasets.py:5-33,
    utils/general.py:9-19): (resized h, w), (top, left), (H2, W2)."""
    size = math.ceil(max(h, w) / stride) * stride
    r = min(size / h, size / w)
    rw, rh = int(round(w * r)), int(round(h * r))
    dw, dh = np.mod(size - rw, 64) / 2, np.mod(size - rh, 64) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return (rh, rw), (top, left), (rh + top + bottom, rw + left + right)


def yolo_preprocess(frames_bgr):
    """detect_faces' BGR2RGB + _preprocess (target_size None) for uint8 frames of one size -> float32 [N, 3, H2, W2] in [0, 1]."""
    out = []
    for img in frames_bgr:
        img = np.ascontiguousarray(img[:, :, ::-1])
        (rh, rw), (top, left), (H2, W2) = yolo_letterbox_geometry(*img.shape[:2])
        canvas = np.full((H2, W2, 3), 114, np.uint8)
        canvas[top:top + rh, left:left + rw] = cv2_resize_linear_u8(img, rw, rh)
        out.append(canvas)
    return torch.from_numpy(np.array(out).transpose(0, 3, 1, 2)).float() / 255.0


def _greedy_nms(boxes, scores, thr):
    """torchvision.ops.nms (PARITY UNPINNED: torchvision absent): descending score, drop when IoU with a kept box > thr; float32."""
    order = np.argsort(-scores, kind='stable')
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep, alive = [], np.ones(len(order), bool)
    for a, i in enumerate(order):
        if not alive[a]:
            continue
        keep.append(int(i))
        rest = order[a + 1:]
        w = np.clip(np.minimum(boxes[i, 2], boxes[rest, 2]) - np.maximum(boxes[i, 0], boxes[rest, 0]), 0, None)
        h = np.clip(np.minimum(boxes[i, 3], boxes[rest, 3]) - np.maximum(boxes[i, 1], boxes[rest, 1]), 0, None)
        inter = w * h
        alive[a + 1:] &= ~((inter / (areas[i] + areas[rest] - inter)) > thr)
    return keep


def yolo_postprocess(pred, net_hw, frame_hw, conf_thres=0.7, iou_thres=0.5, min_face=10):
    """One frame of YoloDetector.detect_faces after the network (face_detector.py:69-104,133-141; general.py:89-165 with one class):
    pred float32 [P, 16] -> int64 [k, 15] rows ``x1 y1 x2 y2 x1 lm x 10`` in frame pixels, or None without a face."""
    f32 = np.float32
    x = pred[pred[:, 4] > f32(conf_thres)].astype(f32)
    conf = x[:, 15] * x[:, 4]
    half_w, half_h = x[:, 2] / f32(2), x[:, 3] / f32(2)
    box = np.stack((x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h), 1)
    sel = conf > f32(conf_thres)
    box, conf, lm = box[sel], conf[sel], x[sel, 5:15]
    if not len(box):
        return None
    keep = _greedy_nms(box, conf, f32(iou_thres))
    box, lm = box[keep], lm[keep]
    H2, W2 = net_hw
    H, W = frame_hw
    gain = min(H2 / H, W2 / W)
    pad = f32((W2 - W * gain) / 2), f32((H2 - H * gain) / 2)
    g = f32(gain)
    lim = np.array([W, H], f32)
    box = (box - np.array([pad[0], pad[1]] * 2, f32)) / g                  # scale_coords: tensor op python scalar = float32 arithmetic
    lm = (lm - np.array([pad[0], pad[1]] * 5, f32)) / g
    box = np.clip(box, 0, np.tile(lim, 2))
    lm = np.clip(lm, 0, np.tile(lim, 5))
    box = (box / np.tile(lim, 2)).astype(np.float64) * np.tile(np.array([W, H], np.float64), 2)      # / gn (float32), * width in double
    lm = (lm / np.tile(lim, 5)).astype(np.float64) * np.tile(np.array([W, H], np.float64), 5)
    box, lm = np.trunc(box).astype(np.int64), np.trunc(lm).astype(np.int64)
    big = ~(box[:, 3] - box[:, 1] < min_face)
    box, lm = box[big], lm[big]
    if not len(box):
        return None
    return np.concatenate((box, box[:, :1], lm), axis=1)


def yolo_prepost_inputs():
    """Frames and crafted predictions of the YoloDetector pre / post-processing golden (shared with the tests): two frame sizes -- 100 x 160
    (longer side a multiple of 32: padding only) and 90 x 150 (resized by 160 / 150, then padded) -- and per size one [2, P, 16] prediction
    tensor: background rows of low objectness, clusters of overlapping boxes whose objectness / class products straddle the thresholds,
    a box across the frame border (clamps), a face lower than min_face, one frame of the second pair without any face."""
    rng = np.random.default_rng(20240517)
    cases = {}
    for tag, (h, w) in (('pad', (100, 160)), ('resize', (90, 150))):
        frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        (rh, rw), (top, left), (H2, W2) = yolo_letterbox_geometry(h, w)
        P = 3 * sum((H2 // s) * (W2 // s) for s in (8, 16, 32))
        pred = np.zeros((2, P, 16), np.float32)
        pred[..., 0] = rng.uniform(0, W2, (2, P))
        pred[..., 1] = rng.uniform(0, H2, (2, P))
        pred[..., 2:4] = rng.uniform(4, 60, (2, P, 2))
        pred[..., 4] = rng.uniform(0.0, 0.6, (2, P))
        pred[..., 5:15] = rng.uniform(0, W2, (2, P, 10))
        pred[..., 15] = rng.uniform(0.0, 1.0, (2, P))
        for n in range(2):
            if tag == 'resize' and n == 1:
                continue                                              # a frame without faces
            centres = [(0.3 * W2, 0.5 * H2, 40.0, 52.0), (0.7 * W2, 0.45 * H2, 30.0, 36.0), (W2 - 6.0, H2 - 20.0, 30.0, 44.0),
                       (0.5 * W2, 0.2 * H2 + top, 9.0, 7.0)]          # (the last: lower than min_face = 10 after truncation)
            rows = rng.choice(P, size=12 * len(centres), replace=False)
            for c, (cx, cy, bw, bh) in enumerate(centres):
                for k in range(12):
                    r = rows[c * 12 + k]
                    jit = rng.normal(0, 2.5, 4)
                    pred[n, r, 0:4] = (cx + jit[0], cy + jit[1], bw + jit[2], bh + jit[3])
                    pred[n, r, 4] = rng.uniform(0.6, 1.0) if k >= 6 else rng.uniform(0.975, 1.0)      # (products straddle 0.7 and 0.97)
                    pred[n, r, 15] = rng.uniform(0.7, 1.0) if k >= 6 else rng.uniform(0.975, 1.0)
                    pred[n, r, 5:15] = np.tile((cx, cy), 5) + rng.normal(0, 0.3 * bw, 10)
        cases[tag] = (frames, pred.astype(np.float32))
    return cases
