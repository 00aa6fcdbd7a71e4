"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU (numpy) restatement of the image <-> tensor converters on the hot path's boundary:

  crops_to_net_input   keep_processor.py:258-259:  img2tensor(face / 255., bgr2rgb=True, float32=True)  (img_util.py:9-37)
                       then torchvision normalize(t, (0.5,)*3, (0.5,)*3)
  net_output_to_bgr_u8 keep_processor.py:272:      tensor2img(t, rgb2bgr=True, min_max=(-1, 1))         (img_util.py:40-94)

Pinned: ``oracle/make_golden.py`` runs the reference's own ``img2tensor`` / ``tensor2img`` (``wm_basicsr/utils/img_util.py``,
imported with ``cv2.cvtColor`` stubbed by the channel flip that COLOR_BGR2RGB / COLOR_RGB2BGR are -- cv2 is not installed in this
image -- and torchvision's ``normalize`` restated as (x - mean) / std) and stores inputs' seeds + outputs in
``tests/golden/ops.npz`` (``conv_in_*``, ``conv_out_*``); ``tests/test_oracle_vs_golden.py`` compares.
"""
import numpy as np


def crops_to_net_input(crops_bgr_u8):
    """list / array of uint8 BGR [H,W,3] crops -> float32 [N,3,H,W] RGB in [-1, 1].
    ``face / 255.`` is a float64 division (img_util.py:23-24 casts float64 -> float32 before the channel swap), then
    (x - 0.5) / 0.5 in float32 (normalize subtracts the mean and divides by the std, in place, in the tensor's dtype)."""
    arr = np.stack([np.asarray(c) for c in crops_bgr_u8], axis=0)
    x = (arr / 255.).astype(np.float32)[..., ::-1]                   # BGR -> RGB
    x = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    return (x - np.float32(0.5)) / np.float32(0.5)


def net_output_to_bgr_u8(frame_rgb_f32):
    """float32 [3,H,W] RGB (unclamped) -> uint8 BGR [H,W,3]: clamp to [-1, 1], (x - min) / (max - min) in float32, RGB -> BGR,
    ``(img * 255.0).round()`` (numpy: half to even) and a truncating cast (img_util.py:66-90)."""
    x = np.clip(np.asarray(frame_rgb_f32, dtype=np.float32), np.float32(-1.0), np.float32(1.0))
    x = (x - np.float32(-1.0)) / np.float32(2.0)
    img = x.transpose(1, 2, 0)[..., ::-1]
    return np.ascontiguousarray((img * 255.0).round().astype(np.uint8))
