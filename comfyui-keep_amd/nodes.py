"""ComfyUI node surface -- drop-in for reference ``nodes.py:17-149``.

Same registry keys / display names (nodes.py:139-149), socket types, widget defaults,
ranges and tooltips (nodes.py:20-31, 50-59, 94-104), ``FUNCTION`` names and ``CATEGORY``.
Error convention of the two processing nodes (nodes.py:65-67, 83-88, 115-117, 131-136):
type-check the pack, ``load_device()`` first, never raise -- log, print the traceback and
return ``(None,)`` -- and always ``offload()`` in ``finally``.  The loader node raises
(keep_model_loader.py:88-95).

What differs from the reference is *behind* ``keep_model.keep_net``: an engine object that
runs ``KEEP.forward`` as hand-written gfx950 kernels (engine/net.py).
"""
import traceback

import torch

from . import logger
from .modules.keep_model_loader import KEEPModelLoader, KEEPModelPack
from .modules.keep_processor import KEEPFaceProcessor
from .modules.utils import KEEP_MODEL_CONFIGS, comfy_image_to_cv2, cv2_to_comfy_image

CATEGORY_NAME = "ComfyUI-KEEP"
DETECTION_MODELS = ['retinaface_resnet50', 'retinaface_mobile0.25', 'YOLOv5l', 'YOLOv5n']

# one process-wide loader -> one model cache per process (reference nodes.py:10-15)
GLOBAL_KEEP_MODEL_LOADER = None


def get_keep_model_loader():
    global GLOBAL_KEEP_MODEL_LOADER
    if GLOBAL_KEEP_MODEL_LOADER is None:
        GLOBAL_KEEP_MODEL_LOADER = KEEPModelLoader()
    return GLOBAL_KEEP_MODEL_LOADER


def _float_widget(tooltip):
    return ("FLOAT", {"default": 1.0, "min": 0.5, "max": 4.0, "step": 0.1, "tooltip": tooltip})


def _bool_widget(default, tooltip):
    return ("BOOLEAN", {"default": default, "tooltip": tooltip})


def _guarded(keep_model, what, body):
    """Shared error/residency convention of the two processing nodes."""
    if not isinstance(keep_model, KEEPModelPack):
        logger.error(f"Invalid KEEP Model Pack provided. Expected KEEPModelPack, got {type(keep_model)}")
        return (None,)
    try:
        keep_model.load_device()
        return (body(KEEPFaceProcessor(keep_model)),)
    except Exception as e:  # noqa: BLE001 -- node contract: never raise into the graph executor
        logger.error(f"{what}: {e}")
        traceback.print_exc()
        return (None,)
    finally:
        keep_model.offload()


class KEEP_ModelLoaderNode:
    _MODEL_TYPES = list(KEEP_MODEL_CONFIGS.keys())

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "model": (s._MODEL_TYPES, {"default": s._MODEL_TYPES[0] if s._MODEL_TYPES else 'KEEP'}),
                "detection_model": (DETECTION_MODELS, {"default": 'retinaface_resnet50'}),
            },
            "optional": {
                "bg_upscale_model": ("UPSCALE_MODEL",),
                "face_upscale_model": ("UPSCALE_MODEL",),
            },
        }

    RETURN_TYPES = ("KEEP_MODEL_PACK",)
    RETURN_NAMES = ("keep_model_pack",)
    FUNCTION = "load_model_pack"
    CATEGORY = CATEGORY_NAME

    def load_model_pack(self, model, detection_model, bg_upscale_model=None, face_upscale_model=None):
        pack = get_keep_model_loader().load_keep_model_pack(
            model_type_str=model, detection_model_str=detection_model,
            bg_upscale_model=bg_upscale_model, face_upscale_model=face_upscale_model)
        return (pack,)


class KEEP_FaceUpscaleImageNode:
    @classmethod
    def INPUT_TYPES(s):
        return {"required": {
            "image": ("IMAGE",),
            "keep_model": ("KEEP_MODEL_PACK",),
            "final_upscale_factor": _float_widget(
                "The final upscaling factor for the output image. The image will be resized to this scale after processing."),
            "has_aligned_face": _bool_widget(False, "Check if the input image is an already aligned 512x512 face."),
            "only_center_face": _bool_widget(
                True, "If the image has multiple faces, only process the one closest to the center."),
            "draw_bounding_box": _bool_widget(
                False, "Draw a bounding box around the detected face on the output image."),
        }}

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "upscale_face_image"
    CATEGORY = CATEGORY_NAME

    def upscale_face_image(self, image: torch.Tensor, keep_model, final_upscale_factor,
                           has_aligned_face, only_center_face, draw_bounding_box):
        def body(processor):
            bgr = comfy_image_to_cv2(image[0].unsqueeze(0))          # first image of the batch only
            out = processor.process_image(
                cv2_image_orig=bgr, final_upscale_factor=final_upscale_factor,
                has_aligned=has_aligned_face, only_center_face=only_center_face, draw_box=draw_bounding_box)
            return cv2_to_comfy_image(out)
        return _guarded(keep_model, "Error processing single image", body)


class KEEP_ProcessImageSequenceNode:
    @classmethod
    def INPUT_TYPES(s):
        return {"required": {
            "images": ("IMAGE",),
            "keep_model": ("KEEP_MODEL_PACK",),
            "final_upscale_factor": _float_widget(
                "The final upscaling factor for the output frames. They will be resized to this scale after processing."),
            "has_aligned_frames": _bool_widget(False, "Check if the input frames are already aligned 512x512 faces."),
            "only_center_face": _bool_widget(
                True, "If frames have multiple faces, only process the one closest to the center."),
            "draw_bounding_box": _bool_widget(
                False, "Draw a bounding box around the detected face on the output frames."),
            "max_clip_length": ("INT", {"default": 20, "min": 1, "max": 100, "step": 1,
                                        "tooltip": "Maximum number of frames to process in a single batch to manage VRAM."}),
        }}

    RETURN_TYPES = ("IMAGE",)
    RETURN_NAMES = ("processed_images",)
    FUNCTION = "process_sequence"
    CATEGORY = CATEGORY_NAME

    def process_sequence(self, images: torch.Tensor, keep_model, final_upscale_factor, has_aligned_frames,
                         only_center_face, draw_bounding_box, max_clip_length):
        def body(processor):
            return processor.process_image_sequence(
                image_sequence_tensor=images, final_upscale_factor=final_upscale_factor,
                has_aligned_frames=has_aligned_frames, only_center_face=only_center_face,
                draw_box=draw_bounding_box, max_clip_length=max_clip_length)
        return _guarded(keep_model, "Error during image sequence processing", body)


NODE_CLASS_MAPPINGS = {
    "KEEP_ModelLoader": KEEP_ModelLoaderNode,
    "KEEP_FaceUpscaleImage": KEEP_FaceUpscaleImageNode,
    "KEEP_ProcessImageSequence": KEEP_ProcessImageSequenceNode,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "KEEP_ModelLoader": "Load KEEP Models",
    "KEEP_FaceUpscaleImage": "KEEP Single Image",
    "KEEP_ProcessImageSequence": "KEEP Image Sequence",
}
