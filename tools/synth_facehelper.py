"""TEST / BENCH INFRASTRUCTURE: a cv2-free stand-in for the reference's ``FaceRestoreHelper``
(wm_facelib/utils/face_restoration_helper.py) with SYNTHETIC face geometry, so that the product's own sequence entry point --
``KEEPFaceProcessor.process_image_sequence`` / ``process_frames_u8`` -- can be driven end to end on a box that has neither the
reference nor OpenCV (the GPU box): every network runs on the engine (RetinaFace, KEEP, ParseNet), every OpenCV-arithmetic step on
the HIP kernels (crop warp, paste-back), and only what the reference does on the host with cv2 is restated here without it:

  * ``get_face_landmarks_5``: the detector IS run (``face_detector.detect_faces`` on the resized frame, exactly the call of
    face_restoration_helper.py:206-221), but a detector with random weights finds no faces, so the landmarks handed on are the
    5-point template carried through a known crop -> frame similarity per (frame, face track);
  * ``estimate_similarity``: the closed-form least-squares similarity (Umeyama, no reflection) in place of
    ``cv2.estimateAffinePartial2D(..., LMEDS)`` (:305) -- identical on exact correspondences;
  * ``get_inverse_affine``: ``cv2.invertAffineTransform`` restated (engine/paste.py:invert_affine) times ``upscale_factor`` (:322-329).
"""
import numpy as np

# FFHQ 5-point template at 512 x 512 (face_restoration_helper.py:75-77: the published facexlib constants)
FACE_TEMPLATE_512 = np.array([[192.98138, 239.94708], [318.90277, 240.1936], [256.63416, 314.01935],
                              [201.26117, 371.41043], [313.08905, 371.15118]], np.float64)


def track_similarity(t, i, H, W, faces):
    """crop -> frame similarity of face track i at frame t: a face spanning ~0.35 of the frame height, drifting slowly."""
    s0 = 0.35 * H / 512.0
    return np.array([[s0, 0.0, (0.25 + 0.25 * i) * W - 256 * s0 + 0.5 * t],
                     [0.0, s0, 0.5 * H - 256 * s0 + 0.2 * t]], np.float64)


def umeyama_similarity(src, dst):
    """Least-squares similarity (rotation + uniform scale + translation, no reflection) mapping src -> dst, as a 2x3 matrix."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    a = (xs * xd).sum()
    b = (xs[:, 0] * xd[:, 1] - xs[:, 1] * xd[:, 0]).sum()
    n = (xs ** 2).sum()
    c, s = a / n, b / n
    R = np.array([[c, -s], [s, c]])
    t = mu_d - R @ mu_s
    return np.concatenate([R, t[:, None]], 1)


class SynthFaceHelper:
    use_parse, pad_blur, is_gray = True, False, False
    face_size = (512, 512)
    det_model = 'retinaface_resnet50'

    def __init__(self, face_detector, face_parse, frame_hw, faces=1, upscale_factor=1):
        self.face_detector, self.face_parse = face_detector, face_parse
        self.H, self.W = frame_hw
        self.n_faces = faces
        self.upscale_factor = upscale_factor
        self.face_template = FACE_TEMPLATE_512.copy()
        self.frame_index = 0
        self.detector_calls = 0
        self.clean_all()

    def begin_sequence(self):
        self.frame_index = 0

    def clean_all(self):
        self.all_landmarks_5, self.det_faces, self.affine_matrices = [], [], []
        self.inverse_affine_matrices, self.cropped_faces, self.restored_faces = [], [], []

    def read_image(self, img):
        self.input_img = img
        self.is_gray = False

    def get_face_landmarks_5(self, only_center_face=False, resize=None, eye_dist_threshold=None, **_):
        # the detector call of :206-221 (the frame resized so that its short side is `resize`; the processor's batched pre-pass
        # replays its stored result through the same attribute)
        if self.face_detector is not None:
            img = self.input_img
            h, w = img.shape[:2]
            if hasattr(self.face_detector, 'engine'):                 # the real per-frame call (not the batched pre-pass's replay)
                if resize is not None and min(h, w) > resize:
                    sc = resize / min(h, w)
                    img = self.resize_for_detector(img, int(w * sc), int(h * sc))
                self.detector_calls += 1
            self.face_detector.detect_faces(img, 0.97)
        t = self.frame_index
        self.frame_index += 1
        n = 1 if only_center_face else self.n_faces
        for i in range(n):
            M = track_similarity(t, i, self.H, self.W, self.n_faces)
            self.all_landmarks_5.append(self.face_template @ M[:, :2].T + M[:, 2])
        return len(self.all_landmarks_5)

    def resize_for_detector(self, img, w, h):
        """The INTER_AREA downscale in front of the detector (:209-212), as an area interpolation on the device (the result stays
        there: the detector input never returns to the host)."""
        import torch
        dev = self.face_detector.engine.device
        x = torch.as_tensor(img).to(dev, non_blocking=True).permute(2, 0, 1)[None].float()
        y = torch.nn.functional.interpolate(x, size=(h, w), mode='area')
        return y.round_().clamp_(0, 255).to(torch.uint8)[0].permute(1, 2, 0).contiguous()

    def estimate_similarity(self, landmarks):
        return umeyama_similarity(landmarks, self.face_template)

    def align_warp_face(self, *a, **kw):
        raise NotImplementedError("SynthFaceHelper: the crop warp runs on the device (keep_warp_affine_u8); enable the GPU cv path")

    def get_inverse_affine(self, save_inverse_affine_path=None):
        from comfyui_keep_amd.engine.paste import invert_affine
        self.inverse_affine_matrices = [invert_affine(M) * self.upscale_factor for M in self.affine_matrices]

    def paste_faces_to_input_image(self, *a, **kw):
        raise NotImplementedError("SynthFaceHelper: the paste-back runs on the device (engine/paste.py); enable the GPU cv path")


def make_processor(net, frame_hw, faces, detector=True):
    """KEEPFaceProcessor over ``net`` with the engine's RetinaFace (resnet50) + ParseNet (synthetic weights) behind a SynthFaceHelper."""
    import types
    import torch
    from comfyui_keep_amd.engine import parsenet as PN
    from comfyui_keep_amd.engine import retinaface as RF
    from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
    dev = net.device
    det = RF.EngineRetinaFace(RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0)).to(dev)) if detector else None
    par = PN.EngineFaceParse(PN.ParseNetEngine(PN.synth_parsenet_state_dict(seed=0)).to(dev))
    helper = SynthFaceHelper(det, par, frame_hw, faces)
    pack = types.SimpleNamespace(keep_net=net, face_helper=helper, bg_upscale_model=None, face_upscale_model=None,
                                 device=torch.device(dev), model_type_str='KEEP')
    proc = KEEPFaceProcessor(pack)
    proc.gpu_paste = True          # (the cv2 self-check cannot run without cv2: the HIP paste is pinned by tests/test_gpu_paste.py)
    return proc, helper
