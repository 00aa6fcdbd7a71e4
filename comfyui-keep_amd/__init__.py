"""comfyui-keep_amd -- MI355X-native drop-in for the KEEP inference hot path of
wildminder/ComfyUI-KEEP (reference __init__.py:1-18): same package-level logger and the
same ``NODE_CLASS_MAPPINGS`` / ``NODE_DISPLAY_NAME_MAPPINGS`` exports.

ComfyUI loads custom-node directories by path (the directory name carries a hyphen, like
the reference's own ``ComfyUI-KEEP``); outside ComfyUI use ``__graft_entry__.load_package()``
which registers this package as ``comfyui_keep_amd``.
"""
import logging
import sys

logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)
if not logger.hasHandlers():
    _h = logging.StreamHandler(sys.stderr)
    _h.setFormatter(logging.Formatter("[%(levelname)s] %(message)s"))
    logger.addHandler(_h)

try:
    from .nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS
except ModuleNotFoundError as _e:  # engine-only use (bench / tests) outside a ComfyUI process
    if _e.name not in ("comfy", "folder_paths"):
        raise
    logger.debug("ComfyUI runtime not present (%s); node surface not registered", _e.name)
    NODE_CLASS_MAPPINGS = {}
    NODE_DISPLAY_NAME_MAPPINGS = {}

__all__ = ['NODE_CLASS_MAPPINGS', 'NODE_DISPLAY_NAME_MAPPINGS']
