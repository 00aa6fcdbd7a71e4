"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *reference* KEEP network (wildminder/ComfyUI-KEEP, mounted read-only at
/root/reference in the build container) so that the CPU restatement in
``oracle/keep_oracle.py`` can be pinned against it and golden vectors generated
(``oracle/make_golden.py``).  /root/reference does not exist on the GPU box, so
nothing under ``tests -m gpu``, ``smoke()`` or ``bench.py`` may import this file.

The reference cannot be imported naively in this image: its package
``__init__`` files pull in cv2 / lmdb / torchvision, and ``keep_arch.py:21`` imports
``diffusers`` which is neither vendored nor installed.  We therefore
  (1) register *empty* namespace packages for ``wm_basicsr`` / ``.archs`` / ``.utils`` /
      ``.ops`` whose ``__path__`` points at the real directories, so sub-module
      files load but the heavyweight package ``__init__``s never run;
  (2) stub ``torchvision`` (only ``__version__`` and ``ops`` are touched at import);
  (3) stub ``diffusers.models.attention`` with the published GEGLU FeedForward
      (diffusers, un-pinned by the reference: ``net.0.proj = Linear(dim, 8*dim)``,
      ``h, g = proj(x).chunk(2, -1); h * gelu(g)``, ``net.2 = Linear(4*dim, dim)``).
      The gate order and exact-erf GELU are pinned by no reference test:
      "parity unpinned" for FeedForward.
No reference source is copied; the stubs below are ours.
"""
import importlib
import logging
import os
import sys
import types

REF_ROOT = os.environ.get("KEEP_REFERENCE_ROOT", "/root/reference")
_DEPS = os.path.join(REF_ROOT, "modules", "deps")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(_DEPS, "wm_basicsr", "archs"))


def _ns(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def _install_stubs():
    import torch
    from torch import nn
    import torch.nn.functional as F

    if "wm_basicsr" in sys.modules and getattr(sys.modules["wm_basicsr"], "_keep_oracle_stub", False):
        return
    base = os.path.join(_DEPS, "wm_basicsr")
    pkg = _ns("wm_basicsr", base)
    pkg._keep_oracle_stub = True
    _ns("wm_basicsr.archs", os.path.join(base, "archs"))
    _ns("wm_basicsr.ops", os.path.join(base, "ops"))
    utils = _ns("wm_basicsr.utils", os.path.join(base, "utils"))

    def get_root_logger(logger_name="basicsr", log_level=logging.INFO, log_file=None):
        return logging.getLogger(logger_name)

    utils.get_root_logger = get_root_logger

    # torchvision: only touched at import time of arch_util.py
    if "torchvision" not in sys.modules:
        tv = _ns("torchvision")
        tv.__version__ = "0.0.0"
        tv.ops = _ns("torchvision.ops")

    # cv2 / torchvision.utils: touched by wm_basicsr/utils/img_util.py (the P4 converters).  The only cv2 call on that path is
    # cvtColor(img, COLOR_BGR2RGB | COLOR_RGB2BGR) on a 3-channel array: the channel flip OpenCV documents for those codes.
    if "cv2" not in sys.modules:
        import numpy as _np
        cv = _ns("cv2")
        cv._keep_oracle_stub = True
        cv.COLOR_BGR2RGB, cv.COLOR_RGB2BGR = 4, 4

        def _cvt(img, code):
            assert code == 4 and img.ndim == 3 and img.shape[2] == 3
            return _np.ascontiguousarray(img[..., ::-1])

        cv.cvtColor = _cvt
    tvu = _ns("torchvision.utils")
    tvu.make_grid = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("make_grid: 4-D tensor2img is not on the path"))
    sys.modules["torchvision"].utils = tvu

    # diffusers.models.attention: FeedForward (GEGLU) + AdaLayerNorm placeholder
    class GEGLU(nn.Module):
        def __init__(self, dim_in, dim_out):
            super().__init__()
            self.proj = nn.Linear(dim_in, dim_out * 2)

        def forward(self, x):
            h, g = self.proj(x).chunk(2, dim=-1)
            return h * F.gelu(g)

    class FeedForward(nn.Module):
        def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", **kw):
            super().__init__()
            assert activation_fn == "geglu"
            inner = int(dim * mult)
            dim_out = dim if dim_out is None else dim_out
            self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out)])

        def forward(self, x):
            for m in self.net:
                x = m(x)
            return x

    class AdaLayerNorm(nn.Module):  # dead branch in the reference (num_embeds_ada_norm=None)
        pass

    _ns("diffusers")
    _ns("diffusers.models")
    att = _ns("diffusers.models.attention")
    att.FeedForward = FeedForward
    att.AdaLayerNorm = AdaLayerNorm


def import_reference_keep():
    """Returns the reference ``KEEP`` nn.Module class (keep_arch.py:860)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    _install_stubs()
    mod = importlib.import_module("wm_basicsr.archs.keep_arch")
    return mod.KEEP


def import_reference_module(name):
    """e.g. 'wm_basicsr.archs.vqgan_arch', 'wm_basicsr.archs.gmflow.gmflow.transformer'."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    _install_stubs()
    return importlib.import_module(name)
