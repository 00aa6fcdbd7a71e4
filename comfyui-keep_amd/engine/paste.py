"""Paste-back compositing on the MI355X: the per-face work of
``FaceRestoreHelper.paste_faces_to_input_image`` (wm_facelib/utils/face_restoration_helper.py:346-475, ``use_parse=True`` and
``use_parse=False``, ``draw_box=False``, colour frames) on HIP kernels (csrc/keep_paste.hip) instead of full-frame cv2 passes on the host:

    parse classes -> mask 0/255 -> 2 x GaussianBlur((101,101), 11) -> 10-px border zeroed -> /255        (:426-437)
    inverse-affine warp of the mask (float weights) and of the restored face (15-bit integer weights)    (:382, :441)
    frame = soft * face + (1 - soft) * frame in float32, face after face; clip, round, uint8             (:463-468)

Only the face's bounding box is touched (outside it the mask is exactly 0 and the blend returns the frame unchanged), so a
1080p frame with 3 faces is 3 boxes of ~(0.75 * face size)^2 pixels instead of ~10 full-frame float passes per face.

OpenCV is not installed in the build image: the arithmetic is restated from OpenCV's published algorithms and checked bit
for bit against the numpy restatement ``oracle/paste_oracle.py`` (tests/test_gpu_paste.py); agreement with cv2 itself is
unmeasured, so the processor uses this path only when ``KEEP_AMD_GPU_PASTE=1``.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hiplib as L

# face parsing classes -> mask value (face_restoration_helper.py:428)
MASK_COLORMAP = (0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 255, 0, 0, 0)
PARSE_BLUR_KSIZE, PARSE_BLUR_SIGMA, PARSE_BORDER = 101, 11.0, 10      # :433-436


def invert_affine(M):
    """cv2.invertAffineTransform (imgwarp.cpp) on a 2x3 matrix, in float64: what cv2.warpAffine does to the matrix it is given
    before it walks the destination image (:382, :441 pass the crop -> frame matrix without WARP_INVERSE_MAP)."""
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = M[1, 1] * D, M[0, 0] * D, -M[0, 1] * D, -M[1, 0] * D
    return np.array([[A11, A12, -A11 * M[0, 2] - A12 * M[1, 2]], [A21, A22, -A21 * M[0, 2] - A22 * M[1, 2]]], np.float64)


_SMALL_GAUSSIAN = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                   7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def gaussian_kernel(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma, CV_32F): fixed tables for ksize <= 7 with sigma <= 0, else exp(-x^2 / 2 sigma^2) in
    double (sigma <= 0: 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8), rounded to float, normalised by the float sum."""
    if sigma <= 0 and ksize in _SMALL_GAUSSIAN:
        return np.array(_SMALL_GAUSSIAN[ksize], np.float32)
    sig = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    cf = np.exp((-0.5 / (sig * sig)) * x * x).astype(np.float32)
    return (cf.astype(np.float64) * (1.0 / float(cf.astype(np.float64).sum()))).astype(np.float32)


def face_box(M_fwd, fw, fh, W, H, margin=2):
    """Bounding box in the frame of the face crop [0,fw] x [0,fh] under the crop -> frame matrix, grown by the bilinear
    footprint + fixed-point slack and clipped to the frame: everything outside it has a zero mask."""
    c = np.array([[0, 0, 1], [fw, 0, 1], [0, fh, 1], [fw, fh, 1]], np.float64) @ np.asarray(M_fwd, np.float64).T
    x0, y0 = math.floor(c[:, 0].min()) - margin, math.floor(c[:, 1].min()) - margin
    x1, y1 = math.ceil(c[:, 0].max()) + margin + 1, math.ceil(c[:, 1].max()) + margin + 1
    return max(0, x0), max(0, y0), min(W, x1), min(H, y1)


ALIGN_BORDER_BGR = (135, 133, 132)          # face_restoration_helper.py:318


def crop_faces(frame_u8, affine_matrices, face_size=(512, 512), device='cuda', border=ALIGN_BORDER_BGR):
    """``align_warp_face`` (face_restoration_helper.py:303-320, border_mode='constant') for one frame: uint8 [H,W,3] frame (numpy
    or tensor) + the frame -> crop similarity matrices of ``cv2.estimateAffinePartial2D`` -> uint8 [F,fh,fw,3] crops on the device
    (``keep_warp_affine_u8``).  None matrices give the reference's black crop (:309-311)."""
    frame = torch.as_tensor(frame_u8).to(device, non_blocking=True).contiguous()
    H, W, _ = frame.shape
    fw, fh = face_size
    out = torch.zeros((len(affine_matrices), fh, fw, 3), dtype=torch.uint8, device=frame.device)
    with torch.cuda.device(frame.device):
        for i, M in enumerate(affine_matrices):
            if M is None:
                continue
            d2s = (C.c_double * 6)(*invert_affine(M).reshape(-1).tolist())
            L.call('keep_warp_affine_u8', frame, H, W, out[i], fh, fw, d2s, int(border[0]), int(border[1]), int(border[2]))
    return out


class GpuPaster:
    """Device buffers are cached per frame size; one instance per processor."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._kern = torch.from_numpy(gaussian_kernel(PARSE_BLUR_KSIZE, PARSE_BLUR_SIGMA)).to(self.device)
        self._lut = torch.tensor(MASK_COLORMAP, dtype=torch.float32, device=self.device)

    def soft_masks(self, parse_classes):
        """parse_classes uint8 [F,h,w] on the device (ParseNet arg-max) -> blurred masks float [F,h,w] on the 0..255 scale;
        border zeroing and /255 happen in the sampler of keep_paste_face."""
        F, h, w = parse_classes.shape
        a = torch.empty((F, h, w), dtype=torch.float32, device=self.device)
        b = torch.empty_like(a)
        tmp = torch.empty_like(a)
        L.call('keep_sep_filter', None, parse_classes, self._lut, tmp, a, F, h, w, self._kern, PARSE_BLUR_KSIZE)
        L.call('keep_sep_filter', a, None, None, tmp, b, F, h, w, self._kern, PARSE_BLUR_KSIZE)
        return b

    def eroded_coverage(self, d2s, H, W, fh, fw, upscale_factor):
        """face_restoration_helper.py:382-391: the warped face square eroded by 2 * upscale, and its area (one float64 sum on the host)
        -> (scratch, inv_mask_erosion, scratch, total_face_area)."""
        a = torch.empty((H, W), dtype=torch.float32, device=self.device)
        b = torch.empty_like(a)
        tmp = torch.empty_like(a)
        L.call('keep_warp_ones', a, H, W, fh, fw, d2s)
        L.call('keep_erode_rect', a, tmp, b, H, W, max(1, int(2 * upscale_factor)))
        total = float(b.sum(dtype=torch.float64).item())
        return a, b, tmp, (1 if total == 0 else total)

    def erosion_mask(self, d2s, H, W, fh, fw, upscale_factor):
        """The ``use_parse=False`` soft mask of one face in FRAME space (face_restoration_helper.py:386-415): coverage of the warped
        face square -> erode(2 * upscale) -> area -> erode(2 * (sqrt(area) // 20)) -> GaussianBlur of the same odd size.  The area
        (one float64 sum) comes back to the host: it sets the two kernel sizes.  None when the blur would need more than 1023 taps (or more than the frame holds)."""
        a, b, tmp, total = self.eroded_coverage(d2s, H, W, fh, fw, upscale_factor)
        w_edge = int(total ** 0.5) // 20
        radius = max(1, w_edge * 2)
        blur = max(1, w_edge * 2)
        if blur % 2 == 0:
            blur += 1
        if blur > 1023 or blur // 2 >= min(H, W):
            return None
        L.call('keep_erode_rect', b, tmp, a, H, W, radius)
        kern = torch.from_numpy(gaussian_kernel(blur, 0)).to(self.device)
        L.call('keep_sep_filter', a, None, None, tmp, b, 1, H, W, kern, blur)
        return b

    def paste(self, frame_u8, faces_u8, inverse_affines, parse_classes=None, upscale_factor=1.0, draw_box=False):
        """frame_u8: uint8 [H,W,3] (numpy or tensor; the background at the output size), faces_u8: uint8 [F,fh,fw,3],
        inverse_affines: F crop -> frame matrices (``get_inverse_affine``; None entries are skipped), parse_classes: uint8
        [F,fh,fw] (``use_parse=True``) or None (``use_parse=False``: the erosion mask above, ``upscale_factor`` as the helper's).
        Returns the composited uint8 [H,W,3] tensor on the device, or None when a face needs a blur wider than the kernels take
        (the caller falls back to the helper)."""
        frame = torch.as_tensor(frame_u8).to(self.device, non_blocking=True).contiguous()
        faces = torch.as_tensor(faces_u8).to(self.device, non_blocking=True).contiguous()
        H, W, _ = frame.shape
        F, fh, fw, _ = faces.shape
        with torch.cuda.device(self.device):
            masks = None
            if parse_classes is not None:
                masks = self.soft_masks(torch.as_tensor(parse_classes).to(self.device, non_blocking=True).contiguous())
            acc = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
            L.call('keep_u8_to_f32', frame, acc, frame.numel())
            for i in range(F):
                M = inverse_affines[i]
                if M is None:
                    continue
                x0, y0, x1, y1 = face_box(M, fw, fh, W, H)
                d2s = (C.c_double * 6)(*invert_affine(M).reshape(-1).tolist())
                if masks is None:
                    fm = self.erosion_mask(d2s, H, W, fh, fw, upscale_factor)
                    if fm is None:
                        return None
                    L.call('keep_paste_face', acc, H, W, faces[i], fm, fh, fw, d2s, x0, y0, x1, y1, -1)
                    continue
                L.call('keep_paste_face', acc, H, W, faces[i], masks[i], fh, fw, d2s, x0, y0, x1, y1, PARSE_BORDER)
            out = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
            L.call('keep_f32_round_u8', acc, out, acc.numel())
            if draw_box:            # :393-400,467-475: green borders over the finished frame, one warped border mask per face
                for i in range(F):
                    M = inverse_affines[i]
                    if M is None:
                        continue
                    d2s = (C.c_double * 6)(*invert_affine(M).reshape(-1).tolist())
                    total = self.eroded_coverage(d2s, H, W, fh, fw, upscale_factor)[3]
                    t = max(1, min(int(1400 / np.sqrt(total)), min(fh, fw) // 20))
                    x0, y0, x1, y1 = face_box(M, fw, fh, W, H)
                    L.call('keep_draw_box', out, H, W, fh, fw, t, d2s, x0, y0, x1, y1)
        return out
