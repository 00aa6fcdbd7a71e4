"""Model pack + loader -- drop-in for reference ``modules/keep_model_loader.py:18-145``.

``KEEPModelPack`` keeps the reference attribute set (keep_net, face_helper, bg_upscale_model,
face_upscale_model, model_type_str, device, offload_device) and the ``load_device()`` /
``offload()`` residency protocol (keep_model_loader.py:28-61).  ``keep_net`` is the
MI355X engine (engine/net.py:KeepNet) instead of the reference ``nn.Module``; it honours the
same ``.to()``, ``.eval()``, ``.load_state_dict(strict=True)``, ``.parameters()`` and
``__call__(x, need_upscale=False)`` contract.

Checkpoint ingestion follows keep_model_loader.py:99-121: ``torch.load(weights_only=True)``,
``params_ema`` preferred over ``params``, legacy key renames ``cross_fuse -> cfa`` and
``fuse_convs_dict -> cft``, strict load, ``.eval()``.  The loader cache is keyed
``(model, detector, bg?, face?)`` and a cache hit returns a NEW pack sharing ``keep_net`` and
``face_helper`` (keep_model_loader.py:74-86).
"""
import os
import sys

import torch
from comfy import model_management

from .. import logger
from .utils import FACELIB_DEST_DIR, FACELIB_MODEL_URLS, KEEP_MODEL_CONFIGS, locate_model_file


def _module_to(obj, device):
    if obj is not None:
        obj.model.to(device)          # spandrel descriptor: the nn.Module is ``.model``


class KEEPModelPack:
    def __init__(self, keep_net, face_helper, bg_upscale_model, face_upscale_model, model_type_str):
        self.keep_net = keep_net
        self.face_helper = face_helper
        self.bg_upscale_model = bg_upscale_model
        self.face_upscale_model = face_upscale_model
        self.model_type_str = model_type_str
        self.device = model_management.get_torch_device()
        self.offload_device = model_management.unet_offload_device()

    def _move_all(self, device):
        if self.keep_net is not None:
            self.keep_net.to(device)
        _module_to(self.bg_upscale_model, device)
        _module_to(self.face_upscale_model, device)
        helper = self.face_helper
        if helper is not None:
            helper.device = device
            for attr in ('face_detector', 'face_parse'):
                obj = getattr(helper, attr, None)
                if obj is None:
                    continue
                if hasattr(obj, 'to'):
                    obj.to(device)
                    continue
                # YoloDetector (detection/yolov5face/face_detector.py:21-46) is a plain class with no .to(): the upstream pack
                # raises AttributeError here as soon as a YOLO detector is selected.  Its network is `.detector` (the engine's
                # EngineYoloModel or the torch Model) and `.device` is what its pre-processing uploads to.
                inner = getattr(obj, 'detector', None)
                if inner is not None and hasattr(inner, 'to'):
                    inner.to(device)
                if hasattr(obj, 'device'):
                    obj.device = device

    def load_device(self):
        """Everything onto the compute device (the engine uploads its packed weight blob)."""
        self._move_all(self.device)

    def offload(self):
        """Everything back to the offload device, then ``soft_empty_cache()``."""
        self._move_all(self.offload_device)
        model_management.soft_empty_cache()


def engine_facelib(helper):
    """put the helper's face-analysis networks on the HIP engine where an engine counterpart exists -- the
    objects keep the reference's call signatures (``face_parse(x)[0]``, ``face_detector.detect_faces(img)``), so
    FaceRestoreHelper itself is unchanged.  ``KEEP_AMD_ENGINE_FACELIB=0`` keeps the torch modules."""
    if os.environ.get('KEEP_AMD_ENGINE_FACELIB', '1') == '0' or helper is None:
        return helper
    fp = getattr(helper, 'face_parse', None)
    if fp is not None and hasattr(fp, 'state_dict') and 'out_mask_conv.conv2d.weight' in fp.state_dict():
        from ..engine.parsenet import EngineFaceParse
        helper.face_parse = EngineFaceParse.from_module(fp)
        logger.debug("face_parse (ParseNet) runs on the HIP engine")
    det = getattr(helper, 'face_detector', None)
    if det is not None and hasattr(det, 'state_dict') and getattr(det, 'backbone', None) in ('Resnet50', 'mobilenet0.25'):
        from ..engine.retinaface import EngineRetinaFace
        helper.face_detector = EngineRetinaFace.from_module(det)            # (the configuration is read off the state dict)
        logger.debug("face_detector (RetinaFace %s) runs on the HIP engine", det.backbone)
    yolo = getattr(det, 'detector', None)                  # YoloDetector (YOLOv5l / YOLOv5n, detection/__init__.py:42-49): its network
    if yolo is not None and hasattr(yolo, 'state_dict') and 'model.0.stem_1.conv.weight' in yolo.state_dict():
        from ..engine.yoloface import EngineYoloModel
        from ..engine.yoloface import yolo_detect_batch
        det.detector = EngineYoloModel.from_module(yolo)    # pre / post-processing stay YoloDetector's own (face_detector.py)
        if hasattr(det, '_preprocess') and hasattr(det, '_postprocess'):      # the processor's batched pre-pass (one network call per chunk)
            import functools
            det.detect_batch = functools.partial(yolo_detect_batch, det)
        logger.debug("face_detector (%s) runs its network on the HIP engine", det.detector.engine.name)
    return helper


def convert_legacy_keys(state_dict):
    """``cross_fuse.* -> cfa.*``, ``fuse_convs_dict.* -> cft.*`` (keep_model_loader.py:110-118)."""
    if not any('cross_fuse' in k or 'fuse_convs_dict' in k for k in state_dict):
        return state_dict
    return {k.replace('cross_fuse', 'cfa').replace('fuse_convs_dict', 'cft'): v for k, v in state_dict.items()}


def select_params(checkpoint):
    """``params_ema`` if present, else ``params``, else the object itself (keep_model_loader.py:107-108)."""
    key = 'params_ema' if 'params_ema' in checkpoint else 'params'
    return checkpoint.get(key, checkpoint)


def _import_face_helper():
    """The face detect/align/paste library is unchanged reference code (out of scope): it is
    imported from the reference's vendored ``modules/deps`` tree, located via KEEP_FACELIB_PATH
    or a ``deps`` directory next to this file."""
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in (os.environ.get('KEEP_FACELIB_PATH'), os.path.join(here, 'deps')):
        if cand and os.path.isdir(cand) and cand not in sys.path:
            sys.path.insert(0, cand)
    from wm_facelib.utils.face_restoration_helper import FaceRestoreHelper
    return FaceRestoreHelper


class KEEPModelLoader:
    def __init__(self):
        self.device = model_management.get_torch_device()
        self.offload_device = model_management.unet_offload_device()
        self.loaded_models = {}

    def _build_net(self, model_type_str):
        from ..engine.net import KeepNet
        entry = KEEP_MODEL_CONFIGS[model_type_str]
        net = KeepNet(**entry['architecture']).to(self.offload_device)
        ckpt_path = locate_model_file(entry['url'], entry['dest_dir'])
        checkpoint = torch.load(ckpt_path, map_location=self.offload_device, weights_only=True)
        net.load_state_dict(convert_legacy_keys(select_params(checkpoint)), strict=True)
        net.eval()
        logger.debug(f"KEEP model '{model_type_str}' loaded onto {self.offload_device}.")
        return net

    def _build_face_helper(self, detection_model_str):
        import folder_paths
        for fname, (url, _sha) in FACELIB_MODEL_URLS.items():
            locate_model_file(url, FACELIB_DEST_DIR, file_name=fname)
        root = os.path.join(folder_paths.models_dir, FACELIB_DEST_DIR)
        os.makedirs(root, exist_ok=True)
        try:
            helper = _import_face_helper()(
                upscale_factor=1, face_size=512, crop_ratio=(1, 1), det_model=detection_model_str,
                save_ext='png', use_parse=True, device=self.offload_device, model_rootpath=root)
        except Exception as e:
            logger.error(f"Error initializing FaceRestoreHelper: {e}")
            raise
        return engine_facelib(helper)

    def load_keep_model_pack(self, model_type_str, detection_model_str,
                             bg_upscale_model=None, face_upscale_model=None):
        cache_key = (model_type_str, detection_model_str,
                     bg_upscale_model is not None, face_upscale_model is not None)
        cached = self.loaded_models.get(cache_key)
        if cached is not None:
            logger.debug(f"Returning cached base models for {cache_key}")
            return KEEPModelPack(cached.keep_net, cached.face_helper,
                                 bg_upscale_model, face_upscale_model, model_type_str)
        if model_type_str not in KEEP_MODEL_CONFIGS:
            raise ValueError(f"Unknown KEEP model type: {model_type_str}")
        net = self._build_net(model_type_str)
        face_helper = self._build_face_helper(detection_model_str)
        self.loaded_models[cache_key] = KEEPModelPack(net, face_helper, None, None, model_type_str)
        return KEEPModelPack(net, face_helper, bg_upscale_model, face_upscale_model, model_type_str)
