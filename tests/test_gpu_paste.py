"""GPU: paste-back compositing kernels (csrc/keep_paste.hip through engine/paste.py) against the numpy restatement
oracle/paste_oracle.py -- bit for bit, on the 1080p / 3-face case of BASELINE configs[3]."""
import os
import sys

import numpy as np
import pytest
import torch

import paste_oracle as P
from comfyui_keep_amd.engine import hiplib as L
from comfyui_keep_amd.engine import paste, synth

pytestmark = pytest.mark.gpu


def test_parse_mask_blur_is_bit_exact():
    _, _, _, classes = synth.synth_paste_case()
    gp = paste.GpuPaster('cuda')
    got = gp.soft_masks(torch.from_numpy(classes).cuda()).cpu().numpy()
    for i in range(classes.shape[0]):
        m = P.MASK_COLORMAP[classes[i].astype(np.int64)]
        ref = P.gaussian_blur(P.gaussian_blur(m, 101, 11), 101, 11)
        assert np.array_equal(got[i], ref), (i, float(np.abs(got[i] - ref).max()))


def test_paste_1080p_3_faces_is_bit_exact():
    frame, faces, mats, classes = synth.synth_paste_case()
    ref = P.paste_faces(frame, list(faces), list(mats), list(classes))
    gp = paste.GpuPaster('cuda')
    got = gp.paste(frame, faces, list(mats), classes).cpu().numpy()
    diff = (got.astype(np.int16) - ref.astype(np.int16))
    assert np.array_equal(got, ref), (int(np.abs(diff).max()), int((diff != 0).sum()))
    assert (got != frame).any()                                   # the faces really landed
    # a skipped face (affine None, :367-368) and a second call on cached buffers
    got2 = gp.paste(frame, faces, [mats[0], None, mats[2]], classes).cpu().numpy()
    ref2 = P.paste_faces(frame, list(faces), [mats[0], None, mats[2]], list(classes))
    assert np.array_equal(got2, ref2)


def test_paste_small_frame_and_upscaled_matrix():
    """720p frame, one face, matrix scaled by upscale_factor = 2 (get_inverse_affine, :332) onto a 1440p canvas."""
    frame, faces, mats, classes = synth.synth_paste_case(H=720, W=1280, n_faces=1)
    big = np.repeat(np.repeat(frame, 2, 0), 2, 1)
    M2 = mats[0] * 2.0
    ref = P.paste_faces(big, [faces[0]], [M2], [classes[0]], upscale_factor=2.0)
    got = paste.GpuPaster('cuda').paste(big, faces[:1], [M2], classes[:1]).cpu().numpy()
    assert np.array_equal(got, ref)


def test_erosion_mask_paste_use_parse_false_is_bit_exact():
    """use_parse=False (face_restoration_helper.py:386-415): warpAffine(ones) -> erode -> area -> erode -> GaussianBlur soft mask in
    frame space, then the same warp + blend: whole 1080p / 3-face composite and each intermediate mask against the oracle."""
    frame, faces, mats, _ = synth.synth_paste_case()
    H, W = frame.shape[:2]
    gp = paste.GpuPaster('cuda')
    import ctypes as C
    for i, M in enumerate(mats):
        d2s = (C.c_double * 6)(*paste.invert_affine(M).reshape(-1).tolist())
        got = gp.erosion_mask(d2s, H, W, 512, 512, 1.0).cpu().numpy()
        ref, _ = P.erosion_soft_mask(M, W, H, 1.0)
        assert np.array_equal(got, ref), (i, float(np.abs(got - ref).max()))
    ref = P.paste_faces(frame, list(faces), list(mats), None, upscale_factor=1.0)
    got = gp.paste(frame, faces, list(mats), None, 1.0).cpu().numpy()
    assert np.array_equal(got, ref), int(np.abs(got.astype(np.int16) - ref.astype(np.int16)).max())
    assert (got != frame).any()
    # a face of ~1330 px in a 1400 x 1440 frame: 2 * (sqrt(area) // 20) + 1 = 133 taps (rounds 2-3 stopped at 127 and fell back)
    rs = np.random.RandomState(4)
    big = rs.randint(0, 256, (1400, 1440, 3)).astype(np.uint8)
    Mb = np.array([[2.6, 0.1, 60.3], [-0.1, 2.6, 70.8]], np.float64)
    face = rs.randint(0, 256, (1, 512, 512, 3)).astype(np.uint8)
    refb = P.paste_faces(big, [face[0]], [Mb], None, upscale_factor=1.0)
    gotb = gp.paste(big, face, [Mb], None, 1.0)
    assert gotb is not None and np.array_equal(gotb.cpu().numpy(), refb)


def test_crop_warp_align_warp_face_is_bit_exact():
    """align_warp_face's cv2.warpAffine(frame, M, (512, 512), borderValue=(135, 133, 132)) (face_restoration_helper.py:316-318)
    on the device: frame -> crop matrices = the inverses of the synthetic case's crop -> frame matrices; the third face hangs over
    the frame edge, so the border colour is exercised; a None matrix gives the reference's black crop."""
    frame, _, mats, _ = synth.synth_paste_case()
    fwd = [P.invert_affine(M) for M in mats]                       # frame -> crop
    got = paste.crop_faces(frame, fwd + [None]).cpu().numpy()
    assert got.shape == (4, 512, 512, 3) and not got[3].any()
    for i, M in enumerate(fwd):
        ref = P.warp_affine_u8(frame, M, 512, 512, border=(135, 133, 132))
        assert np.array_equal(got[i], ref), (i, int(np.abs(got[i].astype(np.int16) - ref.astype(np.int16)).max()))
    assert (got[2] == np.array([135, 133, 132], np.uint8)).all(-1).any()          # the border colour is in the edge face's crop


def test_hip_paste_equals_opencv_where_opencv_exists():
    """The HIP path against cv2 itself (skipped where cv2 is absent, like the build image): the crop warp and the whole
    composite of the 1080p / 3-face case, bit for bit -- the check KEEPFaceProcessor runs once before it switches the GPU
    paste on by default (``_selfcheck_gpu_paste``)."""
    cv2 = pytest.importorskip('cv2')
    from comfyui_keep_amd.modules.keep_processor import opencv_agrees_with_gpu_paste
    assert opencv_agrees_with_gpu_paste('cuda') is True
    frame, _, mats, _ = synth.synth_paste_case()
    fwd = [cv2.invertAffineTransform(M) for M in mats]
    got = paste.crop_faces(frame, fwd).cpu().numpy()
    for i, M in enumerate(fwd):
        assert np.array_equal(got[i], cv2.warpAffine(frame, M, (512, 512), borderMode=cv2.BORDER_CONSTANT, borderValue=(135, 133, 132)))


def test_bad_arguments_fail_loudly():
    a = torch.zeros((8, 8, 3), device='cuda')
    with pytest.raises(L.KeepHipError):
        L.call('keep_sep_filter', None, None, None, a, a, 1, 8, 8, a, 3)      # neither src nor classes


class _StubParse(torch.nn.Module):
    """Stands in for facexlib's ParseNet: logits whose arg-max is a fixed class map per call (one face per call, like
    face_restoration_helper.py:418-427), after checking the input the hook prepared."""
    def __init__(self, faces, classes):
        super().__init__()
        self.faces, self.classes, self.i = faces, classes, 0

    def forward(self, x):
        assert x.shape == (1, 3, 512, 512) and x.dtype == torch.float32
        f = self.faces[self.i].astype(np.float32)
        exp = torch.from_numpy(((f / np.float32(255.)) - np.float32(0.5)) / np.float32(0.5))      # :421-422 on the host
        assert torch.equal(x[0].cpu(), exp.permute(2, 0, 1).flip(0))                               # BGR -> RGB
        logits = torch.nn.functional.one_hot(torch.from_numpy(self.classes[self.i].astype(np.int64)), 19).permute(2, 0, 1)
        self.i += 1
        return [logits[None].float().cuda()]


class _StubHelper:
    use_parse, is_gray, upscale_factor, face_size = True, False, 1, (512, 512)

    def __init__(self, frame, faces, mats, classes):
        self.input_img, self.restored_faces = frame, list(faces)
        self.inverse_affine_matrices = list(mats)
        self.face_parse = _StubParse(faces, classes)
        self.own_calls = 0

    def paste_faces_to_input_image(self, upsample_img=None, draw_box=False, face_upsampler=None):
        self.own_calls += 1
        return upsample_img


def test_processor_paste_hook_runs_on_the_device_when_opted_in():
    import test_host_logic as H   # installs the ComfyUI stubs
    from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
    frame, faces, mats, classes = synth.synth_paste_case()
    helper = _StubHelper(frame, faces, mats, classes)
    pack = KEEPModelPack(None, helper, None, None, 'KEEP')
    pack.device = torch.device('cuda')
    proc = KEEPFaceProcessor(pack)
    assert proc._paste(helper, frame, False) is frame and helper.own_calls == 1        # default: the helper's own method
    proc.gpu_paste = True
    out = proc._paste(helper, frame, False)
    assert helper.own_calls == 1 and helper.face_parse.i == 3
    assert np.array_equal(out, P.paste_faces(frame, list(faces), list(mats), list(classes)))
    helper.face_parse.i = 0
    boxed = proc._paste(helper, frame, True)                                            # draw_box: green borders on the device
    assert helper.own_calls == 1
    ref_boxed = P.paste_faces(frame, list(faces), list(mats), list(classes), draw_box=True)
    assert np.array_equal(boxed, ref_boxed) and int((ref_boxed != out).any(2).sum()) > 500          # (the borders are really there)
    helper.use_parse = False                                                            # erosion-mask path + borders (opt-in path)
    proc._gpu_paste_forced = True
    boxed2 = proc._paste(helper, frame, True)
    assert np.array_equal(boxed2, P.paste_faces(frame, list(faces), list(mats), None, draw_box=True))
    helper.use_parse, proc._gpu_paste_forced = True, False
    # grey faces (is_gray sources: add_restored_face stores [512,512] arrays): pasted as their 3-channel replication (GRAY2BGR)
    grey = [np.ascontiguousarray(f[..., 1]) for f in faces]
    helper2 = _StubHelper(frame, grey, mats, classes)
    helper2.is_gray = True
    helper2.face_parse = _StubParse([np.repeat(g[:, :, None], 3, axis=2) for g in grey], classes)      # the parse input is the replication too
    out_g = proc._paste(helper2, frame, False)
    assert helper2.own_calls == 0
    assert np.array_equal(out_g, P.paste_faces(frame, [np.repeat(g[:, :, None], 3, axis=2) for g in grey], list(mats), list(classes)))


def test_streamed_sequence_equals_the_per_frame_path(gpu_net, monkeypatch):
    """the product's own sequence entry point.  ``process_image_sequence`` with steps 3 + 4 streamed (crops stay on
    the device, finished batch groups handed over by ``run_clips_u8(sink=...)``, ParseNet over up to 32 faces across frames, paste +
    download on a second stream under the next group's forward) against the per-frame path (``KEEP_AMD_STREAM_PASTE=0``: restored
    crops to the host, one frame's faces per ParseNet call, one paste and one download per frame): the same frames BIT FOR BIT --
    float (the node's IMAGE tensor) and uint8 (``process_frames_u8``).  10 frames of 360 x 480 with 2 face tracks -> 20 crops ->
    5 clips of 4, forced into >= 2 batch groups."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import synth_facehelper as SF
    H, W, faces, n = 360, 480, 2, 10
    proc, helper = SF.make_processor(gpu_net, (H, W), faces)
    proc.keep_restored_faces = True          # (off by default: a crop is released once its frame has been composited)
    g = torch.Generator().manual_seed(5)
    frames = torch.rand((n, H, W, 3), generator=g)

    def run(stream, u8=False):
        monkeypatch.setenv('KEEP_AMD_STREAM_PASTE', '1' if stream else '0')
        monkeypatch.setenv('KEEP_AMD_STREAM_GROUPS', '3')
        helper.begin_sequence()
        if u8:
            from comfyui_keep_amd.modules.utils import comfy_image_to_cv2
            return proc.process_frames_u8([comfy_image_to_cv2(frames[i]) for i in range(n)], 1.0, False, False, False, max_clip_length=4)
        return proc.process_image_sequence(frames, 1.0, False, False, False, max_clip_length=4)

    ref = run(False)
    ref_faces = [np.asarray(f) for f in proc.last_restored_faces]
    got = run(True)
    assert type(proc.last_restored_faces).__name__ == '_DeviceFaces' and len(proc.last_restored_faces) == n * faces
    assert got.shape == ref.shape == (n, H, W, 3) and got.dtype == torch.float32
    assert all(np.array_equal(a, b) for a, b in zip(proc.last_restored_faces, ref_faces))          # the restored crops themselves
    assert torch.equal(got, ref)
    assert float((ref - frames).abs().max()) > 0.1                                                # (faces were really pasted)
    u8 = run(True, u8=True)
    assert u8.dtype == torch.uint8 and torch.equal(u8.flip(-1).float() / 255.0, ref)
    ref8 = run(False, u8=True)
    assert all(np.array_equal(u8[i].numpy(), ref8[i]) for i in range(n))
    assert helper.detector_calls == 0                       # the batched detection pre-pass ran (no per-frame detector call)
