"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg are the only callers of anything under oracle/).

CPU restatements (plain PyTorch fp32) of the two face-analysis networks either side of the KEEP hot path:

  * ``parsenet_forward``    wm_facelib/parsing/parsenet.py:72-194 (ConvLayer / ResidualBlock / ParseNet.forward)
  * ``retinaface_forward``  wm_facelib/detection/retinaface/retinaface.py:83-146 (RetinaFace.forward, resnet50 configuration),
                            retinaface_net.py:37-196 (SSH, FPN, heads) and retinaface_utils.py (PriorBox, decode, NMS)

Pinned against the imported reference modules by ``oracle/make_golden_facelib.py`` -> ``tests/golden/facelib.npz``
(``tests/test_oracle_vs_golden.py``).  PARITY UNPINNED for the ResNet-50 trunk of RetinaFace: the reference takes it from
``torchvision.models.resnet50`` (retinaface.py:103-105), torchvision is not vendored under /root/reference and not installed
in the build image; the trunk below restates torchvision's published ResNet-50 v1.5 (Bottleneck with the stride on the 3x3
convolution, state-dict names ``body.conv1 / bn1 / layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1}``) and the golden
generator feeds the SAME restated trunk to the reference's FPN / SSH / heads.
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
