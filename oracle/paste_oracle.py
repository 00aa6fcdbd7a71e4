"""CPU restatement (numpy) of the paste-back compositing of FaceRestoreHelper -- SURVEY.md 8f-2.

TEST INFRASTRUCTURE ONLY (tests/, smoke): the product path is engine/paste.py on the HIP kernels of csrc/keep_paste.hip.

Follows /root/reference/modules/deps/wm_facelib/utils/face_restoration_helper.py:
  get_inverse_affine            :326-333   cv2.invertAffineTransform(M) * upscale_factor
  paste_faces_to_input_image    :346-475   warpAffine(restored face), parse mask -> 2 x GaussianBlur(101, 11) -> border
                                           zeroing -> /255 -> warpAffine -> blend in float32 -> clip, round, uint8
  (the use_parse=False soft mask :386-415: warpAffine(ones) -> erode -> area -> erode -> GaussianBlur, is restated too)

PARITY UNPINNED.  The arithmetic lives in OpenCV (cv2.warpAffine / erode / GaussianBlur / invertAffineTransform), a third-party
dependency that is neither vendored in /root/reference nor installed in this image, so neither this file nor the HIP kernels
could be run against it.  The functions below restate OpenCV 4.x's published algorithms (modules/imgproc/src/imgwarp.cpp:
WarpAffineInvoker + remapBilinear, fixed-point coordinates with AB_BITS = 10 / INTER_BITS = 5 and 15-bit weights for 8-bit
images; filter.dispatch.cpp / smooth.dispatch.cpp: separable Gaussian with BORDER_REFLECT_101, getGaussianKernel; morph.dispatch.cpp:
rectangular erosion with the +inf constant border) from their documentation and source as remembered, with the float
summation ORDER fixed here by definition (OpenCV's SIMD paths may contract or reorder: differences of one float ulp, i.e.
at most one uint8 level on pixels that sit exactly on a rounding boundary).  The GPU tests compare the HIP kernels with THIS
file bit for bit; agreement with cv2 itself is unmeasured.
"""
import numpy as np

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, INTER_TAB_SIZE = 1 << AB_BITS, 1 << INTER_BITS
# face parsing classes -> mask value (face_restoration_helper.py:428)
MASK_COLORMAP = np.array([0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 255, 0, 0, 0], np.float32)
SMALL_GAUSSIAN = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                  7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def invert_affine(M):
    """cv2.invertAffineTransform on a 2x3 float64 matrix (imgwarp.cpp: D = 1/det, zero if singular)."""
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = M[1, 1] * D, M[0, 0] * D, -M[0, 1] * D, -M[1, 0] * D
    b1 = -A11 * M[0, 2] - A12 * M[1, 2]
    b2 = -A21 * M[0, 2] - A22 * M[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]], np.float64)


def inverse_affine_for_paste(affine, upscale_factor):
    """face_restoration_helper.py:331-332."""
    return invert_affine(affine) * upscale_factor


def warp_coords(M_fwd, width, height):
    """Fixed-point source coordinates of cv2.warpAffine(src, M_fwd, (width, height)) without WARP_INVERSE_MAP: the matrix is
    inverted in double, per-column / per-row terms are rounded to 1/1024 px (cvRound = round-half-even), their sum is
    truncated to 1/32 px.  Returns (sx, sy, fx, fy): int32 integer parts and 5-bit fractions, [height, width]."""
    Mi = invert_affine(M_fwd)
    xs = np.arange(width, dtype=np.float64)
    ys = np.arange(height, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * AB_SCALE).astype(np.int64)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * AB_SCALE).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767).astype(np.int32)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767).astype(np.int32)
    return sx, sy, (X & (INTER_TAB_SIZE - 1)).astype(np.int32), (Y & (INTER_TAB_SIZE - 1)).astype(np.int32)


def _fetch(src, sy, sx, border=0):
    """src[sy, sx] with BORDER_CONSTANT `border` (a scalar or one value per channel) outside."""
    h, w = src.shape[:2]
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    v = src[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]
    if v.ndim == 3:
        ok = ok[..., None]
    return np.where(ok, v, np.asarray(border, dtype=v.dtype))


def warp_affine_u8(src, M_fwd, width, height, border=0):
    """cv2.warpAffine(uint8 [h,w,C], M, (width,height)), INTER_LINEAR, BORDER_CONSTANT with borderValue `border` (0 in the
    paste-back, (135, 133, 132) in align_warp_face :316-318): 15-bit integer weights (32-fx)(32-fy)*32 ..., result
    (sum + 2^14) >> 15; a tap outside the source contributes the border colour."""
    sx, sy, fx, fy = warp_coords(M_fwd, width, height)
    w00 = ((32 - fx) * (32 - fy) * 32).astype(np.int64)
    w01 = (fx * (32 - fy) * 32).astype(np.int64)
    w10 = ((32 - fx) * fy * 32).astype(np.int64)
    w11 = (fx * fy * 32).astype(np.int64)
    s = src.astype(np.int64)
    ex = (lambda a: a[..., None]) if src.ndim == 3 else (lambda a: a)
    acc = (_fetch(s, sy, sx, border) * ex(w00) + _fetch(s, sy, sx + 1, border) * ex(w01) + _fetch(s, sy + 1, sx, border) * ex(w10) +
           _fetch(s, sy + 1, sx + 1, border) * ex(w11))
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def warp_affine_f32(src, M_fwd, width, height):
    """cv2.warpAffine(float32 [h,w], ...): float weights of the 1/32-px quantised position, summed left to right in float32
    (each product and each sum rounded: no fused multiply-add)."""
    sx, sy, fx, fy = warp_coords(M_fwd, width, height)
    f32 = np.float32
    ax, ay = fx.astype(f32) / f32(32), fy.astype(f32) / f32(32)
    w00, w01 = (f32(1) - ax) * (f32(1) - ay), ax * (f32(1) - ay)
    w10, w11 = (f32(1) - ax) * ay, ax * ay
    s = src.astype(f32)
    acc = _fetch(s, sy, sx) * w00
    acc = acc + _fetch(s, sy, sx + 1) * w01
    acc = acc + _fetch(s, sy + 1, sx) * w10
    acc = acc + _fetch(s, sy + 1, sx + 1) * w11
    return acc.astype(f32)


def gaussian_kernel(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma, CV_32F): fixed tables for ksize <= 7 with sigma <= 0, else exp(-x^2 / 2 sigma^2)
    evaluated in double, rounded to float, normalised by the float sum."""
    if sigma <= 0 and ksize in SMALL_GAUSSIAN:
        return np.array(SMALL_GAUSSIAN[ksize], np.float32)
    sig = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    scale2x = -0.5 / (sig * sig)
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    cf = np.exp(scale2x * x * x).astype(np.float32)
    s = 1.0 / float(cf.astype(np.float64).sum())
    return (cf.astype(np.float64) * s).astype(np.float32)


def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def sep_filter_f32(img, kern):
    """Row pass then column pass of a separable filter, BORDER_REFLECT_101, float32, taps accumulated in index order
    (acc = acc + k[i] * x[i], each operation rounded)."""
    img = img.astype(np.float32)
    n = len(kern)
    r = n // 2
    h, w = img.shape
    cols = _reflect101(np.arange(-r, w + r), w)
    pad = img[:, cols]
    out = np.zeros_like(img)
    for i in range(n):
        out = out + kern[i] * pad[:, i:i + w]
    rows = _reflect101(np.arange(-r, h + r), h)
    pad = out[rows, :]
    out2 = np.zeros_like(img)
    for i in range(n):
        out2 = out2 + kern[i] * pad[i:i + h, :]
    return out2


def gaussian_blur(img, ksize, sigma):
    return sep_filter_f32(img, gaussian_kernel(ksize, sigma))


def erode_rect(img, k):
    """cv2.erode(img, np.ones((k,k))): anchor at k//2, constant border +inf (pixels outside never win the minimum)."""
    if k <= 1:
        return img.copy()
    a = k // 2
    h, w = img.shape
    big = np.float32(np.inf)
    pad = np.full((h + k, w + k), big, np.float32)
    pad[a:a + h, a:a + w] = img
    out = np.full((h, w), big, np.float32)
    for dy in range(k):
        for dx in range(k):
            out = np.minimum(out, pad[dy:dy + h, dx:dx + w])
    return out


def parse_soft_mask(parse_classes):
    """face_restoration_helper.py:426-436: class map [512,512] -> mask 0/255 -> two GaussianBlur((101,101), 11) -> 10-px
    border zeroed -> /255 (float32)."""
    m = MASK_COLORMAP[parse_classes.astype(np.int64)]
    m = gaussian_blur(m, 101, 11)
    m = gaussian_blur(m, 101, 11)
    t = 10
    m[:t, :] = 0; m[-t:, :] = 0; m[:, :t] = 0; m[:, -t:] = 0
    return (m / np.float32(255.0)).astype(np.float32)


def eroded_coverage(inv_affine, width, height, upscale_factor, face_hw=(512, 512)):
    """:382-391: (inv_mask_erosion, total_face_area)."""
    inv_mask = warp_affine_f32(np.ones(face_hw, np.float32), inv_affine, width, height)
    inv_mask_erosion = erode_rect(inv_mask, int(2 * upscale_factor))
    total = float(np.sum(inv_mask_erosion.astype(np.float64)))      # np.sum over float32: pairwise; magnitude only matters
    return inv_mask_erosion, (1 if total == 0 else total)


def box_thickness(total_face_area, face_hw=(512, 512)):
    """:396-397."""
    t = int(1400 / np.sqrt(total_face_area))
    return max(1, min(t, min(face_hw) // 20))


def border_mask(inv_affine, width, height, thickness, face_hw=(512, 512)):
    """:393-400: the warped ``mask_border`` (its three channels are equal: one is computed) -> bool [height,width], > 0.5 (:470)."""
    h, w = face_hw
    m = np.ones((h, w), np.float32)
    m[thickness:h - thickness, thickness:w - thickness] = 0            # cv2.rectangle((t, t), (w - t - 1, h - t - 1), 0, -1): corners inclusive
    return warp_affine_f32(m, inv_affine, width, height) > np.float32(0.5)


def erosion_soft_mask(inv_affine, width, height, upscale_factor, face_hw=(512, 512)):
    """:386-415 (use_parse=False): returns (inv_soft_mask [height,width] float32, total_face_area)."""
    inv_mask_erosion, total = eroded_coverage(inv_affine, width, height, upscale_factor, face_hw)
    w_edge = int(total ** 0.5) // 20
    erosion_radius = max(1, w_edge * 2)
    center = erode_rect(inv_mask_erosion, erosion_radius)
    blur = max(1, w_edge * 2)
    if blur % 2 == 0:
        blur += 1
    return gaussian_blur(center, blur, 0), total


def paste_faces(upsample_img, restored_faces, inverse_affines, parse_classes=None, upscale_factor=1.0, draw_box=False):
    """paste_faces_to_input_image(upsample_img=..., draw_box=..., face_upsampler=None) for colour frames.
    upsample_img uint8 [H,W,3] (already at the output size), restored_faces: uint8 [512,512,3] each, inverse_affines: the
    matrices of get_inverse_affine, parse_classes: per face the ParseNet arg-max map [512,512] (use_parse=True) or None."""
    h_up, w_up = upsample_img.shape[:2]
    up = upsample_img
    borders = []
    for idx, face in enumerate(restored_faces):
        M = inverse_affines[idx]
        if M is None:
            continue
        if draw_box:
            _, total = eroded_coverage(M, w_up, h_up, upscale_factor, face.shape[:2])
            borders.append(border_mask(M, w_up, h_up, box_thickness(total, face.shape[:2]), face.shape[:2]))
        inv_restored = warp_affine_u8(face, M, w_up, h_up)
        if parse_classes is not None:
            soft = warp_affine_f32(parse_soft_mask(parse_classes[idx]), M, w_up, h_up)
        else:
            soft, _ = erosion_soft_mask(M, w_up, h_up, upscale_factor, face.shape[:2])
        soft = soft[:, :, None].astype(np.float32)
        # :463  float32 arithmetic, every product / sum rounded (numpy promotes uint8 with float32 to float32)
        up = soft * inv_restored.astype(np.float32) + (np.float32(1) - soft) * up.astype(np.float32)
    if np.issubdtype(up.dtype, np.floating):
        up = np.clip(up, 0, 255)
    out = np.round(up).astype(np.uint8)
    for b in borders:                                    # :467-475
        out[b] = np.array([0, 255, 0], np.uint8)
    return out
