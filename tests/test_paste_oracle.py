"""CPU: the paste-back restatement (oracle/paste_oracle.py) -- structural properties + its regression vectors.
(cv2 is absent from the image: the restatement is unpinned against OpenCV itself; see its header.)"""
import hashlib
import os

import numpy as np
import pytest

import paste_oracle as P
from conftest import GOLDEN
from comfyui_keep_amd.engine import paste as host_paste
from comfyui_keep_amd.engine import synth


def test_identity_warp_and_inverse():
    img = (np.random.RandomState(0).rand(64, 80, 3) * 255).astype(np.uint8)
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    assert np.array_equal(P.warp_affine_u8(img, eye, 80, 64), img)
    f = np.random.RandomState(1).rand(64, 80).astype(np.float32)
    assert np.array_equal(P.warp_affine_f32(f, eye, 80, 64), f)
    # a pure integer shift moves pixels exactly; what leaves the image reads the constant border 0
    sh = np.array([[1, 0, 5], [0, 1, -3]], np.float64)
    out = P.warp_affine_u8(img, sh, 80, 64)
    assert np.array_equal(out[:61, 5:], img[3:, :75]) and not out[:, :5].any() and not out[61:].any()
    A = np.array([[0.8, -0.3, 100.5], [0.3, 0.8, 40.25]])
    assert np.abs(P.invert_affine(P.invert_affine(A)) - A).max() < 1e-12
    assert np.array_equal(P.inverse_affine_for_paste(A, 2.0), P.invert_affine(A) * 2.0)


def test_filters():
    k = P.gaussian_kernel(101, 11)
    assert len(k) == 101 and abs(float(k.astype(np.float64).sum()) - 1) < 1e-6 and np.array_equal(k, k[::-1])
    assert np.array_equal(P.gaussian_kernel(5, 0), np.array([0.0625, 0.25, 0.375, 0.25, 0.0625], np.float32))
    const = np.full((40, 50), 3.0, np.float32)
    assert np.abs(P.gaussian_blur(const, 11, 2.0) - 3.0).max() < 1e-5           # reflect-101 borders keep constants
    m = np.ones((9, 9), np.float32); m[4, 4] = 0
    e = P.erode_rect(m, 3)
    assert e[3:6, 3:6].max() == 0 and e[0, 0] == 1 and e.sum() == 81 - 9       # +inf border: edges do not erode
    e2 = P.erode_rect(m, 2)                                                     # anchor k//2 = 1: window [-1, 0]
    assert e2[4, 4] == 0 and e2[5, 5] == 0 and e2[3, 3] == 1


def test_host_helpers_equal_the_restatement():
    """engine/paste.py carries its own copies of the two host-side formulas (the product never imports oracle/)."""
    A = np.array([[0.45, -0.09, 300.5], [0.09, 0.45, 200.25]])
    assert np.array_equal(host_paste.invert_affine(A), P.invert_affine(A))
    assert np.array_equal(host_paste.gaussian_kernel(101, 11.0), P.gaussian_kernel(101, 11))
    assert tuple(host_paste.MASK_COLORMAP) == tuple(int(v) for v in P.MASK_COLORMAP)
    x0, y0, x1, y1 = host_paste.face_box(A, 512, 512, 1920, 1080)
    soft = P.warp_affine_f32(np.ones((512, 512), np.float32), A, 1920, 1080)
    ys, xs = np.nonzero(soft)
    assert x0 <= xs.min() and xs.max() < x1 and y0 <= ys.min() and ys.max() < y1


def test_paste_regression_vectors():
    g = np.load(os.path.join(GOLDEN, 'paste_1080p_3faces.npz'))
    frame, faces, mats, classes = synth.synth_paste_case()
    soft0 = P.parse_soft_mask(classes[0])
    assert abs(float(soft0.astype(np.float64).sum()) - float(g['soft0_sum'])) < 1e-3 and soft0[256, 256] == g['soft0_center']
    out = P.paste_faces(frame, list(faces), list(mats), list(classes))
    assert np.array_equal(out[::8, ::8], g['decimated'])
    assert np.array_equal(np.frombuffer(hashlib.sha1(out.tobytes()).digest(), np.uint8), g['sha1'])
    # a zero mask leaves the frame untouched; pixels outside every face are bit-identical to the background
    zero = P.paste_faces(frame, list(faces), list(mats), [np.zeros_like(c) for c in classes])
    assert np.array_equal(zero, frame)
    assert int((out != frame).any(-1).sum()) == int(g['changed_pixels'])


def test_draw_box_paints_a_closed_frame_around_each_face():
    """draw_box (face_restoration_helper.py:393-400,467-475): an identity-placed face gets a green frame of the rule's thickness
    (int(1400 / sqrt(area)), at least 1, at most face / 20) along the face square and nothing else changes."""
    bg = np.full((600, 640, 3), 90, np.uint8)
    face = np.full((512, 512, 3), 200, np.uint8)
    M = np.array([[1, 0, 40], [0, 1, 30]], np.float64)                       # crop -> frame: a pure shift
    cls = np.full((512, 512), 1, np.uint8)
    plain = P.paste_faces(bg, [face], [M], [cls])
    boxed = P.paste_faces(bg, [face], [M], [cls], draw_box=True)
    diff = (plain != boxed).any(2)
    green = (boxed == np.array([0, 255, 0], np.uint8)).all(2)
    assert diff.sum() > 0 and np.array_equal(diff, green & diff) and not green[:30].any() and not green[:, :40].any()
    t = P.box_thickness(P.eroded_coverage(M, 640, 600, 1.0)[1])
    assert t == max(1, min(int(1400 / np.sqrt(510.0 * 510.0)), 25))           # the 2 x 2 erosion takes a pixel off the 512 x 512 coverage
    ys, xs = np.where(green)
    assert (ys.min(), xs.min(), ys.max(), xs.max()) == (30, 40, 30 + 511, 40 + 511)
    inner = green[30 + t + 1:30 + 511 - t, 40 + t + 1:40 + 511 - t]
    assert not inner.any() and green[30:30 + t, 40:40 + 512].all() and green[30:30 + 512, 40 + 512 - t:40 + 512].all()


def test_restatement_equals_opencv_where_opencv_exists():
    """THE PIN (runs wherever cv2 is installed; the build image has none -> skipped there): every OpenCV call the paste-back
    and the crop warp make -- warpAffine uint8 with and without a border colour, warpAffine float32, GaussianBlur((101,101), 11),
    invertAffineTransform -- bit for bit against oracle/paste_oracle.py on the synthetic 1080p / 3-face case."""
    cv2 = pytest.importorskip('cv2')
    frame, faces, mats, classes = synth.synth_paste_case()
    H, W = frame.shape[:2]
    for i, M in enumerate(mats):
        assert np.array_equal(cv2.invertAffineTransform(M), P.invert_affine(M)), i
        assert np.array_equal(cv2.warpAffine(faces[i], M, (W, H)), P.warp_affine_u8(faces[i], M, W, H)), i
        fwd = cv2.invertAffineTransform(M)
        assert np.array_equal(cv2.warpAffine(frame, fwd, (512, 512), borderMode=cv2.BORDER_CONSTANT, borderValue=(135, 133, 132)),
                              P.warp_affine_u8(frame, fwd, 512, 512, border=(135, 133, 132))), i
        m = P.MASK_COLORMAP[classes[i].astype(np.int64)]
        blur = cv2.GaussianBlur(cv2.GaussianBlur(m, (101, 101), 11), (101, 101), 11)
        assert np.array_equal(blur, P.gaussian_blur(P.gaussian_blur(m, 101, 11), 101, 11)), i
        soft = P.parse_soft_mask(classes[i])
        assert np.array_equal(cv2.warpAffine(soft, M, (W, H)), P.warp_affine_f32(soft, M, W, H)), i
        # use_parse=False chain (:386-415): coverage, erosions (even and odd kernels), sigma-0 Gaussian of a computed size
        cov = cv2.warpAffine(np.ones((512, 512), np.float32), M, (W, H))
        assert np.array_equal(cov, P.warp_affine_f32(np.ones((512, 512), np.float32), M, W, H)), i
        for k in (2, 9, 24):
            assert np.array_equal(cv2.erode(cov, np.ones((k, k), np.uint8)), P.erode_rect(cov, k)), (i, k)
        for ks in (5, 7, 13, 25):
            assert np.array_equal(cv2.GaussianBlur(cov, (ks, ks), 0), P.gaussian_blur(cov, ks, 0)), (i, ks)
