"""Deterministic synthetic weights and clips (no network, no checkpoints on the GPU box).

Everything is generated from a counter-based integer hash (splitmix64) so that the
build container (where the reference is imported to make golden vectors) and the GPU
box (where only this package exists) produce bit-identical fp32 tensors.

All-zero tensors of the reference's default init (every CFT conv ``keep_arch.py:459-463``,
every CFA linear ``keep_arch.py:510-517``, ``position_emb`` ``keep_arch.py:928``, all
biases) are given non-zero values here, otherwise parity tests would be vacuous for
those modules ("hard parts").
"""
import hashlib
import math

import numpy as np
import torch

from .arch import DEFAULT_ARCH, state_dict_spec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
HEAD_GAIN = 0.125       # see _sigma_for('generator.blocks.24.weight')


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _name_key(name: str, seed: int) -> np.uint64:
    h = hashlib.blake2b(f'{seed}:{name}'.encode(), digest_size=8).digest()
    return np.uint64(int.from_bytes(h, 'little'))


def uniform_pm1(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in (-1, 1), a pure function of (name, seed, index)."""
    key = _name_key(name, seed)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        bits = _splitmix64(_splitmix64(idx ^ key) + key)
    top = (bits >> np.uint64(40)).astype(np.float64)          # 24 random bits
    return (top + 0.5) * (2.0 / 16777216.0) - 1.0


def _sigma_for(name: str, shape):
    """(mean, std) of the synthetic distribution for a non-normalisation tensor."""
    if name == 'position_emb':
        return 0.0, 0.3
    if name == 'quantize.embedding.weight':
        return 0.0, 0.7
    if name == 'idx_pred_layer.1.weight':                 # peaked logits (top-1/top-2 margins ~ trained nets)
        return 0.0, 4.0 / math.sqrt(shape[1])
    if name == 'generator.blocks.24.weight':
        # the output head (VQ:339-343 last conv, 64 -> 3): scaled so that the synthetic net's restored frames live in the
        # range a trained net's do (~[-1, 1]: tensor2img clamps to it, img_util.py:66-90) -- that is the range the <= 1e-3
        # max-abs tolerance of the parity tests is stated for, and the range the HQ encoder sees through prev_out
        fan_in = shape[1] * shape[2] * shape[3]
        return 0.0, HEAD_GAIN / math.sqrt(fan_in)
    if len(shape) == 4:                                   # conv weight
        fan_in = shape[1] * shape[2] * shape[3]
        return 0.0, 1.0 / math.sqrt(fan_in)
    if len(shape) == 2:                                   # linear weight
        return 0.0, 1.0 / math.sqrt(shape[1])
    return 0.0, 0.05                                      # conv / linear bias


def _is_norm_affine(name, spec):
    """1-D 'weight' tensors are always normalisation scales (conv/linear weights are >=2-D);
    a 1-D 'bias' is a norm shift iff its sibling 'weight' is 1-D."""
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'weight':
        return len(spec[name]) == 1
    if leaf == 'bias':
        sib = name[:-4] + 'weight'
        return sib in spec and len(spec[sib]) == 1
    return False


# GMFlow in a PHYSICAL regime (round 4).  With i.i.d. random weights the global matching (softmax over all 4096 positions of
# f0(p) . f1(q), gmflow/matching.py:6-40) is decided by the position-independent part of the features: every pixel "matches" the same
# few attractor positions and the flows are hundreds of pixels on a 512-pixel frame (round 3: median 57 px, max 419 px) -- the
# warp -> hq_encoder -> Kalman path was validated on samples from outside the image.  Three changes to the SYNTHETIC flownet make the
# matching find the actual (sub-pixel, see synth_clip) motion, measured on the imported reference (oracle/make_golden.py prints the
# statistics): |flow| median 0.9 px, p99 5.7 px, 0.6 % of the pixels above 8 px:
#   * the backbone's last 1x1 convolution has zero-sum rows (no constant feature component: its input is post-ReLU, all channel means
#     positive) and a gain of 1.5 (softmax temperature: sub-pixel flows come from the soft arg-max over neighbouring positions);
#   * the transformer's LayerNorm scales / shifts are 0.05 of the usual ones: its random attention updates perturb the features
#     instead of replacing them (both stay non-zero: every multiply / add of the layer is still exercised).
FLOW_NORM_GAIN = 0.05
FLOW_FEATURE_GAIN = 1.5


def _flownet_override(name, shape, v):
    """float64 values of a synthetic ``flownet.*`` tensor -> the values that are used (see the comment above)."""
    if name.startswith('flownet.model.transformer.') and name.rsplit('.', 2)[-2] in ('norm1', 'norm2'):
        return v * FLOW_NORM_GAIN
    if name == 'flownet.model.backbone.conv2.weight':
        w = v.reshape(shape[0], -1)
        return ((w - w.mean(axis=1, keepdims=True)) * FLOW_FEATURE_GAIN).reshape(-1)
    if name == 'flownet.model.backbone.conv2.bias':
        return v * FLOW_NORM_GAIN
    return v


def synth_state_dict(cfg=None, seed: int = 0, flow_regime: str = 'physical'):
    """name -> fp32 CPU tensor for every entry of ``state_dict_spec(cfg)``.  ``flow_regime='wide'``: i.i.d. flownet weights
    (round 3's net: flows of hundreds of pixels -- the out-of-range edge case of the warp path)."""
    assert flow_regime in ('physical', 'wide'), flow_regime
    spec = state_dict_spec(cfg)
    out = {}
    for name, shape in spec.items():
        n = int(np.prod(shape))
        u = uniform_pm1(name, n, seed)
        leaf = name.rsplit('.', 1)[-1]
        if len(shape) == 1 and _is_norm_affine(name, spec):
            mean, std = (1.0, 0.1) if leaf == 'weight' else (0.0, 0.1)
        else:
            mean, std = _sigma_for(name, shape)
        v = mean + u * (math.sqrt(3.0) * std)
        if name.startswith('flownet.') and flow_regime == 'physical':
            v = _flownet_override(name, shape, v)
        out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return out


CLIP_MOTION = (0.5, 0.3)        # pixels per frame (x, y): consecutive frames differ by a sub-pixel translation


def synth_clip(T: int = 20, B: int = 1, size: int = 512, seed: int = 1234, phase: float = 0.0, pattern: str = 'texture'):
    """a textured pattern that TRANSLATES slowly + 0.2 % hash noise, fp32 [B,T,3,S,S] in [-1,1].
    (``pattern='waves'``: round 3's clip, kept with ``flow_regime='wide'`` weights as the out-of-range edge case.)

    Three octaves of value noise (hash lattices of 9 / 17 / 33-pixel cells, smoothstep-interpolated, amplitudes 0.5 / 0.6 /
    0.5, an own lattice per clip and channel) sampled at (x - 0.5 t - phase, y - 0.3 t - 0.7 phase): every local patch is unique,
    so a flow network can match it, and the motion is 0.58 px per frame -- the regime of a face crop in a video (round 3's clip was
    two plane waves + 10 % per-frame noise: no patch could be matched).  Element-wise float64 arithmetic only (bit-identical on
    every machine: the reference's goldens are generated in another container than the one the GPU tests run in)."""
    if pattern == 'waves':
        return _waves_clip(T, B, size, seed, phase)
    assert pattern == 'texture', pattern
    S = size
    cells, amps = (9, 17, 33), (0.5, 0.6, 0.5)
    ys, xs = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64), indexing='ij')
    out = np.empty((B, T, 3, S, S), dtype=np.float32)
    for b in range(B):
        tabs = []
        for oi, c in enumerate(cells):
            n = S // c + 8
            tabs.append(np.stack([uniform_pm1(f'tex:{b}:{oi}:{ch}', n * n, seed).reshape(n, n) for ch in range(3)]))
        for t in range(T):
            X = xs - (CLIP_MOTION[0] * t + phase)
            Y = ys - (CLIP_MOTION[1] * t + 0.7 * phase)
            img = np.zeros((3, S, S))
            for oi, c in enumerate(cells):
                gx, gy = X / c + 3.0, Y / c + 3.0                    # (+3 cells: the lattice index stays >= 0 for 20 frames)
                ix, iy = np.floor(gx), np.floor(gy)
                fx, fy = gx - ix, gy - iy
                fx, fy = fx * fx * (3.0 - 2.0 * fx), fy * fy * (3.0 - 2.0 * fy)
                ix, iy = ix.astype(np.int64), iy.astype(np.int64)
                tb = tabs[oi]
                top = tb[:, iy, ix] * (1.0 - fx) + tb[:, iy, ix + 1] * fx
                bot = tb[:, iy + 1, ix] * (1.0 - fx) + tb[:, iy + 1, ix + 1] * fx
                img += amps[oi] * (top * (1.0 - fy) + bot * fy)
            for ch in range(3):
                u = uniform_pm1(f'clip:{b}:{t}:{ch}', S * S, seed).reshape(S, S)
                out[b, t, ch] = np.clip(img[ch] + 0.002 * u, -1.0, 1.0).astype(np.float32)
    return torch.from_numpy(out)


def _waves_clip(T, B, size, seed, phase):
    """Round 3's clip (``synth_clip(pattern='waves')``): two plane waves + 10 % per-frame hash noise, fp32 [B,T,3,S,S] in [-1,1].

    x[t,c,y,x] = 0.6 sin(2pi(3x+2y)/S + 0.7c + 0.15t + phase) + 0.3 sin(2pi(5x-4y)/S + 0.05t) + 0.1 u
    """
    S = size
    ys, xs = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64), indexing='ij')
    out = np.empty((B, T, 3, S, S), dtype=np.float32)
    for b in range(B):
        for t in range(T):
            for c in range(3):
                base = (0.6 * np.sin(2 * np.pi * (3 * xs + 2 * ys) / S + 0.7 * c + 0.15 * t + phase + 0.9 * b)
                        + 0.3 * np.sin(2 * np.pi * (5 * xs - 4 * ys) / S + 0.05 * t))
                u = uniform_pm1(f'clip:{b}:{t}:{c}', S * S, seed).reshape(S, S)
                out[b, t, c] = (base + 0.1 * u).astype(np.float32)
    return torch.from_numpy(out)


def ramp_image(h: int = 512, w: int = 512):
    """uint8 BGR ramp, pixel(y,x,c) = (37y + 91x + 53c) mod 256."""
    y, x, c = np.meshgrid(np.arange(h), np.arange(w), np.arange(3), indexing='ij')
    return ((37 * y + 91 * x + 53 * c) % 256).astype(np.uint8)


def synth_paste_case(H: int = 1080, W: int = 1920, n_faces: int = 3, face: int = 512, seed: int = 7):
    """Deterministic paste-back workload (BASELINE configs[3]: a 1080p frame with 3 faces): a ramp frame,
    ``n_faces`` restored crops, their crop -> frame matrices (rotated / scaled similarities as ``get_inverse_affine`` returns
    them; the last face hangs over the frame edge) and ParseNet-like class maps (skin / brow / nose ellipses inside hair and
    background).  numpy arrays: frame uint8 [H,W,3], faces uint8 [n,face,face,3], matrices float64 [n,2,3], classes uint8 [n,face,face]."""
    yy, xx = np.mgrid[0:H, 0:W]
    frame = np.stack([(xx * 3 + yy * 5) % 256, (xx * 7 + yy * 2 + 40) % 256, (xx + yy * 11 + 80) % 256], -1).astype(np.uint8)
    fy, fx = np.mgrid[0:face, 0:face].astype(np.float64)
    faces, mats, classes = [], [], []
    for i in range(n_faces):
        noise = uniform_pm1(f'paste.face{i}', face * face * 3, seed).reshape(face, face, 3)
        base = np.stack([128 + 100 * np.sin(fx / (23.0 + i) + c) * np.cos(fy / (31.0 - i) - c) for c in range(3)], -1)
        faces.append(np.clip(np.rint(base + 20 * noise), 0, 255).astype(np.uint8))
        s = (0.45, 0.30, 0.62, 0.5)[i % 4]
        th = (-0.2, 0.15, 0.05, 0.3)[i % 4]
        cx = (W * 0.3, W * 0.62, W - 60.0, W * 0.5)[i % 4]          # third face: partly outside the frame
        cy = (H * 0.35, H * 0.6, H * 0.25, H * 0.5)[i % 4]
        a, b = s * math.cos(th), s * math.sin(th)
        mats.append(np.array([[a, -b, cx - (a - b) * face / 2 + 0.37 * i], [b, a, cy - (a + b) * face / 2 - 0.21 * i]], np.float64))
        r = np.hypot((fx - face / 2) / (0.34 * face), (fy - face * 0.52) / (0.42 * face))
        cls = np.zeros((face, face), np.uint8)
        cls[r < 1.15] = 17                                            # hair
        cls[r < 0.95] = 1                                             # skin
        cls[np.hypot((fx - face / 2) / 40.0, (fy - face * 0.55) / 60.0) < 1] = 10   # nose
        cls[(np.abs(fy - face * 0.38) < 8) & (np.abs(np.abs(fx - face / 2) - 70) < 40)] = 2   # brows
        cls[fy > face * 0.9] = 14                                     # neck
        classes.append(cls)
    return frame, np.stack(faces), np.stack(mats), np.stack(classes)
