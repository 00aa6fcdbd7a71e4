"""Config table + image<->tensor converters (reference ``modules/utils.py:41-90,155-166`` and
``wm_basicsr/utils/img_util.py:9-94``), restated without the cv2 dependency: the only cv2
calls on this path are channel swaps (``cvtColor`` RGB<->BGR), which are index flips.

Numerics that change results:
  * ``comfy_image_to_cv2`` TRUNCATES ``(x*255).astype(uint8)``   (utils.py:155-160)
  * ``tensor2img`` clamps to [-1,1], maps ``(x+1)/2*255`` and ROUNDS half-to-even (numpy
    ``.round()``), then RGB->BGR                                  (img_util.py:66-90)
  * crops enter the net as ``float32(u8/255.) -> (x-0.5)/0.5``    (keep_processor.py:258-259)
"""
import os

import numpy as np
import torch

from ..engine.arch import DEFAULT_ARCH

_COMMON = {k: DEFAULT_ARCH[k] for k in (
    'img_size', 'emb_dim', 'dim_embd', 'n_head', 'n_layers', 'codebook_size', 'kalman_attn_head_dim',
    'num_uncertainty_layers', 'cfa_list', 'cfa_nhead', 'cfa_dim', 'cond', 'nf', 'ch_mult',
    'attn_resolutions', 'res_blocks', 'quantizer_type', 'beta')}

# defaults the reference back-fills into every entry (utils.py:76-90)
_DEFAULT_ARCH_PARAMS = {
    'gumbel_straight_through': False, 'gumbel_kl_weight': 1e-8, 'vqgan_path': None, 'latent_size': 256,
    'fix_modules': ['quantize', 'generator'], 'flownet_path': None, 'cfa_nlayers': 4,
    'cross_residual': True, 'mask_ratio': 0.,
}

_RELEASE = 'https://github.com/jnjaby/KEEP/releases/download/v1.0.0/'

KEEP_MODEL_CONFIGS = {
    'KEEP': {
        'architecture': dict(_COMMON, cft_list=['16', '32', '64'], temp_reg_list=['32'], **_DEFAULT_ARCH_PARAMS),
        'url': _RELEASE + 'KEEP-b76feb75.pth',
        'dest_dir': 'keep_models/KEEP',
    },
    'Asian': {
        'architecture': dict(_COMMON, cft_list=['32', '64', '128', '256'], temp_reg_list=[], **_DEFAULT_ARCH_PARAMS),
        'url': _RELEASE + 'KEEP_Asian-4765ebe0.pth',
        'dest_dir': 'keep_models/KEEP',
    },
}

FACELIB_MODEL_URLS = {
    'detection_Resnet50_Final.pth': (_RELEASE + 'detection_Resnet50_Final.pth', None),
    'detection_mobilenet0.25_Final.pth': (_RELEASE + 'detection_mobilenet0.25_Final.pth', None),
    'yolov5n-face.pth': (_RELEASE + 'yolov5n-face.pth', None),
    'yolov5l-face.pth': (_RELEASE + 'yolov5l-face.pth', None),
    'parsing_parsenet.pth': (_RELEASE + 'parsing_parsenet.pth', None),
}
FACELIB_DEST_DIR = 'facedetection'


def locate_model_file(url, model_dir_name, file_name=None):
    """Resolve ``ComfyUI/models/<model_dir_name>/<file>`` (reference utils.py:101-153).

    Downloading is plumbing that stays with ComfyUI / the reference helper (out of scope,
    SURVEY.md section 2); this build only resolves the path and fails with a clear message
    when the checkpoint is not already on disk (there is no network on the build/GPU box).
    """
    import folder_paths  # ComfyUI runtime
    file_name = file_name or os.path.basename(url).split('?')[0]
    path = os.path.join(folder_paths.models_dir, model_dir_name, file_name)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{file_name} not found at {path}. Download it from {url} (the reference node does this "
            f"via torch.hub) and place it there.")
    return path


# --------------------------------------------------------------------------- converters
def comfy_image_to_cv2(comfy_image: torch.Tensor) -> np.ndarray:
    """IMAGE [1,H,W,3] (or [H,W,3]) float RGB 0..1 -> uint8 BGR [H,W,3]; truncating cast."""
    if comfy_image.ndim == 3:
        comfy_image = comfy_image.unsqueeze(0)
    rgb = (comfy_image.cpu().numpy().squeeze(0) * 255).astype(np.uint8)
    return np.ascontiguousarray(rgb[..., ::-1])


def cv2_to_comfy_image(cv2_image: np.ndarray) -> torch.Tensor:
    """uint8 BGR [H,W,3] -> IMAGE [1,H,W,3] float32 RGB 0..1."""
    rgb = np.ascontiguousarray(cv2_image[..., ::-1])
    return torch.from_numpy(rgb.astype(np.float32) / 255.0).unsqueeze(0)


def crops_to_net_input(crops_bgr_u8) -> torch.Tensor:
    """list of uint8 BGR [512,512,3] crops -> fp32 [N,3,512,512] RGB in [-1,1].

    ``img2tensor(face / 255., bgr2rgb=True, float32=True)`` then ``normalize(0.5, 0.5)``
    (keep_processor.py:258-259): the division is float64, cast to float32, then (x-0.5)/0.5.
    """
    arr = np.stack([np.asarray(c) for c in crops_bgr_u8], axis=0)            # [N,H,W,3] BGR u8
    x = (arr / 255.).astype(np.float32)[..., ::-1]                            # RGB
    t = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))
    return t.sub_(0.5).div_(0.5)


def net_output_to_bgr_u8(t: torch.Tensor) -> np.ndarray:
    """fp32 [3,H,W] RGB (unclamped) -> uint8 BGR [H,W,3]; ``tensor2img(rgb2bgr=True, min_max=(-1,1))``."""
    x = t.detach().float().cpu().clamp_(-1, 1)
    x = (x - (-1)) / (1 - (-1))
    img = x.numpy().transpose(1, 2, 0)[..., ::-1]
    return np.ascontiguousarray((img * 255.0).round().astype(np.uint8))
