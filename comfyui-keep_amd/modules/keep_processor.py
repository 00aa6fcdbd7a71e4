"""Frame / clip driver -- drop-in for reference ``modules/keep_processor.py:117-307``.

Hot loop #1 of the path: crops of all frames are stacked frame-major
(faces of one frame adjacent), cut into chunks of ``max_clip_length`` and every chunk is one
independent ``keep_net`` clip (keep_processor.py:256-270).  Behaviour kept bug-compatible:
  * multi-face sequences interleave faces inside a clip (P1);
  * a chunk of length 1 is duplicated to T=2 and frame 0 kept (keep_processor.py:266-268, 173-175);
  * with ``has_aligned_frames`` the restored faces are computed and then NOT used: the output
    is the resized input (keep_processor.py:289-291) -- see ``last_restored_faces`` below;
  * ComfyUI progress: N + N + N ticks, then one tick per frame that had faces (KP:200-304).
Detection / tracking / alignment / paste-back stay with the reference's ``FaceRestoreHelper``
(out of scope, unchanged); tracking + smoothing (KP:33-115, 216-231) are restated in
``face_tracks.py`` because they live in the reference's own glue file.

What this build adds: independent clips are handed to the engine in one call
(``keep_net.run_clips_u8``) so it can batch equal-length clips per GPU (bounded by free HBM) and,
when a torch.distributed group or the worker pool (``KEEP_AMD_GPUS``, engine/pool.py) is up, shard them across
GPUs.  Clips share no state and the kernel plans are batch-invariant (tiles / split-K follow the per-image geometry,
DESIGN.md section 6), so the result equals the sequential one-clip-at-a-time loop BIT FOR BIT
(tests/test_gpu_net.py::test_config3_config4_clip_mixes_equal_sequential).
The paste-back compositing can run on the GPU (``KEEP_AMD_GPU_PASTE``, ``_paste`` below).

Round 5 -- the sequence path STREAMS steps 3 and 4 (``_restore_and_paste_streamed``): with the engine's net, its ParseNet and the
GPU paste available, crops that were warped on the GPU stay there, the clip loop hands every finished batch group over on the
device (``run_clips_u8(sink=...)``), ParseNet runs on batches of up to 32 faces ACROSS frames (the reference: one face per call,
face_restoration_helper.py:418-424), and the frames whose faces are all restored are composited and downloaded on a second HIP
stream while the next group is being restored.  Same arithmetic, same order per frame: the output equals the per-frame path bit
for bit (tests/test_gpu_paste.py::test_streamed_sequence_equals_the_per_frame_path); every other configuration keeps that path.
"""
import os

import numpy as np
import torch
from tqdm import tqdm

from .utils import comfy_image_to_cv2, crops_to_net_input, cv2_to_comfy_image, net_output_to_bgr_u8
from .face_tracks import smooth_center_face, smooth_tracked_faces

try:  # ComfyUI runtime
    from comfy.utils import ProgressBar, tiled_scale
except ModuleNotFoundError:  # engine-only use outside ComfyUI
    tiled_scale = None

    class ProgressBar:  # minimal stand-in with the same calls
        def __init__(self, total):
            self.total, self.current = total, 0

        def update(self, n):
            self.current += n


def _cv2():
    import cv2
    return cv2


def _resize(img, w, h, interp_name):
    if img.shape[0] == h and img.shape[1] == w:
        return img
    cv2 = _cv2()
    return cv2.resize(img, (w, h), interpolation=getattr(cv2, interp_name))


def _is_gray(img, threshold=10):
    """wm_facelib/utils/misc.py:146-160: mean variance of channel differences <= threshold."""
    if img.ndim == 2:
        return True
    c = img.astype(np.int16)
    d = ((c[..., 0] - c[..., 1]).var() + (c[..., 1] - c[..., 2]).var() + (c[..., 2] - c[..., 0]).var()) / 3.0
    return bool(d <= threshold)


def opencv_agrees_with_gpu_paste(device):
    """True iff the HIP paste-back (engine/paste.py) reproduces OpenCV bit for bit on this installation: the composite
    of engine/synth.py's 1080p / 3-face case computed with cv2 calls in the order of face_restoration_helper.py:382,426-468
    against ``GpuPaster.paste``, and the :316-318 crop warp against ``crop_faces``.  Raises ImportError without cv2."""
    import cv2
    from ..engine import paste as gp
    from ..engine import synth
    frame, faces, mats, classes = synth.synth_paste_case()
    H, W = frame.shape[:2]
    up = frame
    lut = np.asarray(gp.MASK_COLORMAP, np.float32)
    for face, M, cls in zip(faces, mats, classes):
        inv_restored = cv2.warpAffine(face, M, (W, H))
        m = lut[cls.astype(np.int64)]
        m = cv2.GaussianBlur(cv2.GaussianBlur(m, (101, 101), 11), (101, 101), 11)
        t = gp.PARSE_BORDER
        m[:t, :] = 0; m[-t:, :] = 0; m[:, :t] = 0; m[:, -t:] = 0
        soft = cv2.warpAffine(m / np.float32(255.0), M, (W, H))[:, :, None]
        up = soft * inv_restored + (1 - soft) * up
    ref = np.round(np.clip(up, 0, 255)).astype(np.uint8)
    got = gp.GpuPaster(device).paste(frame, faces, list(mats), torch.from_numpy(classes)).cpu().numpy()
    if not np.array_equal(got, ref):
        return False
    fwd = cv2.invertAffineTransform(mats[2])
    crop = cv2.warpAffine(frame, fwd, (512, 512), borderMode=cv2.BORDER_CONSTANT, borderValue=gp.ALIGN_BORDER_BGR)
    return bool(np.array_equal(gp.crop_faces(frame, [fwd], device=device)[0].cpu().numpy(), crop))


class _ReplayDetector:
    """Stands in for ``face_detector`` while the helper post-processes ONE frame of a batched detection pass: returns the
    stored ``detect_faces`` result of that frame."""

    def __init__(self, result):
        self.result = result

    def detect_faces(self, image, conf_threshold=0.8, *args, **kwargs):
        return self.result


class _DeviceFaces:
    """``last_restored_faces`` of the streamed path: the restored crops stay where they were produced (the GPU, or host memory for a
    pool worker's) and are fetched on access -- nothing downloads 236 MB per 300 crops for a list only tests and tools read."""

    def __init__(self, items):
        self._items = list(items)

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self._items)))]
        t = self._items[i]
        return np.ascontiguousarray(t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t))

    def __iter__(self):
        return (self[i] for i in range(len(self._items)))


_EMITTED = object()   # placeholder of a restored crop whose frame has been composited (streamed path)
_PINNED_U8 = {}      # one cached pinned frame buffer per shape (hipHostMalloc of 0.8 GB costs ~0.1 s; frames never outlive the node call)


class _ConvertedFrames:
    """The node's IMAGE batch as uint8 BGR frames -- ``comfy_image_to_cv2`` (utils.py:155-160) for the whole sequence, on the device
    (``keep_comfy_to_bgr_u8``: x * 255 in float32, truncating cast, RGB -> BGR) instead of three numpy passes per frame on one host
    core: a worker thread uploads the float frames in chunks (pageable -> device), converts them and downloads the uint8 frames into
    ONE pinned buffer; ``frames[i]`` / ``frames[a:b]`` wait only for the chunks they touch, so the detection pre-pass of the first
    frames runs under the conversion of the rest.  Items are numpy views of the pinned buffer (what the helper and cv2 take)."""

    def __init__(self, seq, device, chunk_bytes=128 << 20):
        import threading
        self.n, self.H, self.W = int(seq.shape[0]), int(seq.shape[1]), int(seq.shape[2])
        self._seq, self._dev = seq, device
        per = self.H * self.W * 3
        self._chunk = max(1, min(self.n, chunk_bytes // (per * 4)))
        self._nchunks = -(-self.n // self._chunk)
        self._done = [threading.Event() for _ in range(self._nchunks)]
        self._ready = threading.Event()          # the pinned buffer exists
        self._err = None
        self._arr = None
        self._thread = threading.Thread(target=self._run, name='keep-comfy-convert', daemon=True)
        self._thread.start()

    def _run(self):
        try:
            from ..engine import hiplib as L
            shape = (self.n, self.H, self.W, 3)
            buf = _PINNED_U8.get(shape)
            if buf is None:
                _PINNED_U8.clear()
                try:
                    buf = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
                    _PINNED_U8[shape] = buf
                except RuntimeError:                                  # more than the host will pin
                    buf = torch.empty(shape, dtype=torch.uint8)
            self._buf, self._arr = buf, buf.numpy()
            self._ready.set()
            with torch.cuda.device(self._dev):
                st = torch.cuda.Stream(device=self._dev)
                pend = None
                with torch.cuda.stream(st):
                    for c in range(self._nchunks):
                        a, b = c * self._chunk, min(self.n, (c + 1) * self._chunk)
                        src = self._seq[a:b]
                        if not src.is_contiguous():
                            src = src.contiguous()
                        d = src.to(self._dev, non_blocking=True)      # (a pageable source: the call returns when the chunk is staged)
                        u = torch.empty((b - a, self.H, self.W, 3), dtype=torch.uint8, device=self._dev)
                        L.call('keep_comfy_to_bgr_u8', d, u, (b - a) * self.H * self.W)
                        buf[a:b].copy_(u, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(st)
                        if pend is not None:                           # chunk c - 1 finished under this chunk's upload
                            pend[1].synchronize()
                            self._done[pend[0]].set()
                        pend = (c, ev)
                    pend[1].synchronize()
                    self._done[pend[0]].set()
        except BaseException as e:                                     # (out of memory, a lost device ...): the host converter, loudly
            import logging
            logging.getLogger('ComfyUI-KEEP').warning("device-side IMAGE conversion failed (%s); converting on the host", e)
            try:
                self._arr = np.stack([comfy_image_to_cv2(self._seq[i].unsqueeze(0)) for i in range(self.n)])
            except BaseException as e2:                                # surfaced by the first access
                self._err = e2
            self._ready.set()
            for d in self._done:
                d.set()

    def _wait(self, a, b):
        self._ready.wait()
        for c in range(a // self._chunk, min(self._nchunks, -(-b // self._chunk))):
            self._done[c].wait()
        if self._err is not None:
            raise self._err

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            a, b, step = i.indices(self.n)
            if step != 1:
                return [self[j] for j in range(a, b, step)]
            if b <= a:
                return []
            self._wait(a, b)
            return [self._arr[j] for j in range(a, b)]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        self._wait(i, i + 1)
        return self._arr[i]

    def __iter__(self):
        return (self[i] for i in range(self.n))


def frames_from_comfy(seq, device):
    """IMAGE batch [N,H,W,3] float32 (host or device) -> N uint8 BGR frames.  On the device when there is one and the library loads
    (``_ConvertedFrames``); ``KEEP_AMD_DEVICE_CONVERT=0``, another dtype / layout, or no GPU: the per-frame host converter."""
    dev = torch.device(device)
    if (os.environ.get('KEEP_AMD_DEVICE_CONVERT', '1') != '0' and dev.type == 'cuda' and torch.cuda.is_available()
            and isinstance(seq, torch.Tensor) and seq.dtype == torch.float32 and seq.dim() == 4 and seq.shape[-1] == 3
            and seq.shape[0] > 0 and seq.shape[1] > 0 and seq.shape[2] > 0):
        try:
            from ..engine import hiplib as L
            L.load()
            if dev.index is None:
                dev = torch.device('cuda', torch.cuda.current_device())
            return _ConvertedFrames(seq, dev)
        except Exception:
            pass
    return [comfy_image_to_cv2(seq[i].unsqueeze(0)) for i in range(seq.shape[0])]


def split_clips(num_faces: int, max_clip_length: int):
    """[(start, end)] chunk boundaries of the flat crop list (keep_processor.py:263-264)."""
    return [(s, min(s + max_clip_length, num_faces)) for s in range(0, num_faces, max_clip_length)]


class KEEPFaceProcessor:
    def __init__(self, model_pack):
        self.keep_net = model_pack.keep_net
        self.face_helper = model_pack.face_helper
        self.bg_upscale_model = model_pack.bg_upscale_model
        self.face_upscale_model = model_pack.face_upscale_model
        self.device = model_pack.device
        self.model_type_str = model_pack.model_type_str
        # restored 512x512 faces of the last call, uint8 BGR (the reference discards them on the
        # aligned-sequence path; exposed so callers / tests can read what the net produced)
        self.last_restored_faces = []
        # streamed sequence path: keep the restored crops (on the GPU) in ``last_restored_faces`` after the call -- off by default
        self.keep_restored_faces = os.environ.get('KEEP_AMD_KEEP_RESTORED', '0') == '1'
        # reference quirk P2: `KEEP Image Sequence` with has_aligned=True restores every frame
        # and then returns the *input* frames.  Default: bug-compatible.  KEEP_AMD_RETURN_RESTORED_ALIGNED=1 (or setting
        # the attribute) returns the restored faces instead.
        self.return_restored_aligned = os.environ.get('KEEP_AMD_RETURN_RESTORED_ALIGNED', '0') == '1'
        # paste-back compositing on the GPU (engine/paste.py).  Opt-in: its arithmetic restates OpenCV's and
        # could not be checked against cv2 itself in the build image.
        # KEEP_AMD_GPU_PASTE: '1' on, '0' off; unset = 'auto': the first paste runs ``opencv_agrees_with_gpu_paste`` -- the HIP
        # path against cv2 itself on a synthetic 1080p / 3-face frame, bit for bit -- and switches the GPU path on only if that
        # holds on this installation (cv2 is a hard dependency of the reference helper, so it exists wherever this code pastes).
        env = os.environ.get('KEEP_AMD_GPU_PASTE')
        self.gpu_paste = {'1': True, '0': False}.get(env, None)
        self._gpu_paste_forced = env == '1'        # the erosion-mask branch (use_parse=False) is not part of the cv2 self-check
        self._paster = None

    # ------------------------------------------------------------------ net invocation
    def _restore_clips(self, crops_tensor, max_clip_length):
        """crops_tensor [1,N,3,512,512] on device -> [N,3,512,512] restored (fp32, unclamped)."""
        n = crops_tensor.shape[1]
        clips = []
        for s, e in split_clips(n, max_clip_length):
            clip = crops_tensor[:, s:e]
            if clip.shape[1] == 1 and not getattr(self.keep_net, 'supports_single_frame', False):
                clip = torch.cat([clip, clip], dim=1)    # the reference net needs T>=2 (KP:173-178); frame 0 is kept
            clips.append(clip)
        run_clips = getattr(self.keep_net, 'run_clips', None)
        if run_clips is not None:                        # engine: batch / shard independent clips
            outs = run_clips(clips, need_upscale=False)
        else:
            outs = [self.keep_net(c, need_upscale=False)
                    for c in tqdm(clips, desc="Restoring faces with KEEP")]
        kept = []
        for (s, e), o in zip(split_clips(n, max_clip_length), outs):
            kept.append(o[:, 0:1] if e - s == 1 else o)
        return torch.cat(kept, dim=1).squeeze(0)

    def _restore_crops_u8(self, crops, max_clip_length):
        """list of uint8 BGR [512,512,3] crops -> list of restored uint8 BGR crops (same order).

        Same chunking as ``_restore_clips`` (keep_processor.py:263-270); when the net offers ``run_clips_u8`` the
        uint8<->fp32 conversions of keep_processor.py:258-259,272-273 run on the GPU and only uint8 crosses PCIe,
        otherwise the host converters are used (reference behaviour)."""
        run_u8 = getattr(self.keep_net, 'run_clips_u8', None)
        if run_u8 is None:
            x = crops_to_net_input(crops).unsqueeze(0).to(self.device)
            return [net_output_to_bgr_u8(f) for f in self._restore_clips(x, max_clip_length)]
        crops = [np.asarray(c) for c in crops]
        spans = split_clips(len(crops), max_clip_length)
        clips = []
        t1_ok = getattr(self.keep_net, 'supports_single_frame', False)
        for s, e in spans:
            clip = crops[s:e]                            # a clip = a list of crops: the engine copies them straight into its
            if e - s == 1 and not t1_ok:                 # pinned upload buffer (no stacked intermediate)
                clip = clip + clip                       # the reference net needs T>=2 (KP:173-178): duplicate, keep frame 0
            clips.append(clip)                           # (the engine restores a lone frame as T=1: same frame 0, half the work)
        outs = run_u8(clips)
        if outs is None:          # a non-root rank of a run sharded over several GPUs: rank 0 holds the frames and pastes
            return None
        faces = []
        for (s, e), o in zip(spans, outs):
            o = o.numpy()
            faces.extend(o[k] for k in range(e - s))
        return [np.ascontiguousarray(f) for f in faces]

    def _run_upscaler(self, model, cv2_image):
        if model is None:
            return cv2_image
        t = cv2_to_comfy_image(cv2_image).to(self.device).movedim(-1, -3)
        s = tiled_scale(t, lambda a: model.model(a), tile_x=512, tile_y=512, overlap=64,
                        upscale_amount=model.scale)
        return comfy_image_to_cv2(torch.clamp(s.movedim(-3, -1), min=0, max=1.0))

    def _final_background(self, frame_bgr, factor):
        up = self._run_upscaler(self.bg_upscale_model, frame_bgr)
        h, w, _ = frame_bgr.shape
        return _resize(up, int(w * factor), int(h * factor), 'INTER_LANCZOS4')

    # ------------------------------------------------------------------ single image
    @torch.no_grad()
    def process_image(self, cv2_image_orig: np.ndarray, final_upscale_factor: float, has_aligned: bool,
                      only_center_face: bool, draw_box: bool):
        helper = self.face_helper
        helper.upscale_factor = final_upscale_factor
        bg_img_final = self._final_background(cv2_image_orig, final_upscale_factor)

        if has_aligned:
            face = _resize(cv2_image_orig, 512, 512, 'INTER_LINEAR')
            helper.is_gray = _is_gray(face, threshold=10)
            helper.cropped_faces = [face]
            crops = [face]
        else:
            helper.clean_all()
            helper.read_image(cv2_image_orig)
            if helper.get_face_landmarks_5(only_center_face=only_center_face, resize=640,
                                           eye_dist_threshold=5) == 0:
                return bg_img_final
            self._align_warp(helper)
            crops = list(helper.cropped_faces)
            if not crops:
                return bg_img_final

        # one face -> T=2 duplicate, keep frame 0; several faces -> one "clip" of T=#faces (KP:173-178)
        faces = self._restore_crops_u8(crops, max_clip_length=max(len(crops), 1))
        if faces is None:
            return None
        self.last_restored_faces = faces
        helper.restored_faces = [f.astype('uint8') for f in faces]

        if not has_aligned:
            helper.get_inverse_affine(None)
            out = self._paste(helper, bg_img_final, draw_box)
        else:
            out = helper.restored_faces[0]
            if self.face_upscale_model:
                out = self._run_upscaler(self.face_upscale_model, out)
            side = int(512 * final_upscale_factor)
            out = _resize(out, side, side, 'INTER_LANCZOS4')
        return out if out is not None else bg_img_final

    # ------------------------------------------------------------------ paste-back
    def _paste(self, helper, bg, draw_box):
        """``helper.paste_faces_to_input_image(upsample_img=bg, draw_box=..., face_upsampler=...)`` behind one call
        (face_restoration_helper.py:346-475).  With ``gpu_paste`` and the configuration the loader builds (use_parse=True,
        colour frame already at the output size, no box, no face upsampler, every crop at the helper's face size) the
        per-face masks, warps and the blend run on the MI355X (engine/paste.py); every other configuration -- and every
        helper that is not the reference's -- goes to the helper's own method."""
        if self._gpu_cv_path() and self._gpu_paste_applies(helper, bg, draw_box):
            return self._paste_gpu(helper, bg, draw_box)
        return helper.paste_faces_to_input_image(upsample_img=bg, draw_box=draw_box, face_upsampler=self.face_upscale_model)

    def _gpu_cv_path(self):
        """Whether the OpenCV-arithmetic kernels (paste-back, crop warp) may stand in for cv2: forced by KEEP_AMD_GPU_PASTE, else
        decided once per processor by comparing them with cv2 itself on this installation."""
        if self.gpu_paste is None:
            try:
                self.gpu_paste = bool(opencv_agrees_with_gpu_paste(self.device))
            except Exception:                            # no cv2 / no GPU / any surprise: the helper's own path
                self.gpu_paste = False
        return self.gpu_paste

    def _align_warp(self, helper, keep_on_device=False):
        """``helper.align_warp_face()`` (face_restoration_helper.py:256-320).  With the GPU cv path and the default configuration
        (no pad_blur, constant border) only the similarity fit stays on the host (``cv2.estimateAffinePartial2D``, :305); the
        ``cv2.warpAffine`` that produces the 512 x 512 crops (:316-318) runs on the device (``keep_warp_affine_u8``)."""
        if (not self._gpu_cv_path() or getattr(helper, 'pad_blur', False) or not hasattr(helper, 'face_template')
                or getattr(helper.input_img, 'dtype', None) != np.uint8):       # (16-bit sources are float64 after read_image)
            return helper.align_warp_face()
        from ..engine.paste import crop_faces
        fit = getattr(helper, 'estimate_similarity', None)          # (a helper may bring its own fit: bench.py's cv2-free stand-in)
        cv2 = _cv2() if fit is None else None
        mats = [fit(lm) if fit is not None else cv2.estimateAffinePartial2D(lm, helper.face_template, method=cv2.LMEDS)[0]
                for lm in helper.all_landmarks_5]
        crops = crop_faces(helper.input_img, mats, tuple(helper.face_size), self.device)
        helper.affine_matrices.extend(mats)
        if keep_on_device:                                          # the streamed sequence path: no PCIe round trip of the crops
            helper.cropped_faces.extend(crops[i] for i in range(len(mats)))
        else:
            crops = crops.cpu().numpy()
            helper.cropped_faces.extend(crops[i] for i in range(len(mats)))

    def _gpu_paste_applies(self, helper, bg, draw_box):
        faces, mats = getattr(helper, 'restored_faces', None), getattr(helper, 'inverse_affine_matrices', None)
        if self.face_upscale_model is not None:            # (draw_box is on the device since round 4: keep_draw_box)
            return False
        use_parse = getattr(helper, 'use_parse', False)
        if not use_parse and not getattr(self, '_gpu_paste_forced', False):
            # face_restoration_helper.py:386-415 (erode + GaussianBlur(k, 0) at data-dependent sizes): bit-equal to
            # oracle/paste_oracle.py, but ``opencv_agrees_with_gpu_paste`` compares only the parse-mask composite and the crop warp
            # with cv2 itself -- this branch stays opt-in (KEEP_AMD_GPU_PASTE=1) until it has been compared on an installation
            return False
        if (use_parse and getattr(helper, 'face_parse', None) is None) or not faces or mats is None:
            return False
        if len(faces) != len(mats) or not isinstance(bg, np.ndarray) or bg.dtype != np.uint8 or bg.ndim != 3 or bg.shape[2] != 3:
            return False
        h, w = getattr(helper, 'input_img', bg).shape[:2]
        up = getattr(helper, 'upscale_factor', 1)
        if (int(h * up), int(w * up)) != bg.shape[:2]:          # :355-356 would resize the background first
            return False
        fw, fh = getattr(helper, 'face_size', (512, 512))
        # grey sources (is_gray: add_restored_face stored bgr2gray + AdaIN faces, [512,512]): the reference replicates the channel
        # BEFORE the warp and the parse input (cv2.cvtColor GRAY2BGR, :377-378,418-419), and so does _paste_gpu
        return (fh, fw) == (512, 512) and all(np.asarray(f).shape in ((512, 512, 3), (512, 512)) and np.asarray(f).dtype == np.uint8
                                              for f in faces)

    @torch.no_grad()
    def _paste_gpu(self, helper, bg, draw_box=False):
        from ..engine import hiplib as L
        from ..engine.paste import GpuPaster
        if self._paster is None:
            self._paster = GpuPaster(self.device)
        faces = torch.from_numpy(np.ascontiguousarray(np.stack(
            [np.asarray(f) if np.asarray(f).ndim == 3 else np.repeat(np.asarray(f)[:, :, None], 3, axis=2)      # GRAY2BGR
             for f in helper.restored_faces]))).to(self.device)
        if not getattr(helper, 'use_parse', False):       # :386-415 erosion mask instead of the parse mask
            out = self._paster.paste(bg, faces, list(helper.inverse_affine_matrices), None, getattr(helper, 'upscale_factor', 1), draw_box)
            if out is None:                                # a face larger than the blur kernel takes: the helper's own path
                return helper.paste_faces_to_input_image(upsample_img=bg, draw_box=draw_box, face_upsampler=None)
            return out.cpu().numpy()
        # :418-424  BGR uint8 -> RGB float (x/255 - 0.5)/0.5, one face per ParseNet call like the reference
        x = torch.empty(faces.shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.call('keep_img2tensor', faces, x, faces.numel() // 3)
        engine = getattr(helper.face_parse, 'engine', None)
        if engine is not None:            # ParseNet on the HIP engine (engine/parsenet.py): all faces of the frame in one batch
            classes = engine.classes(x)
        else:
            classes = []
            for i in range(faces.shape[0]):
                logits = helper.face_parse(x[i:i + 1].permute(0, 3, 1, 2).contiguous())[0]
                classes.append(logits.argmax(dim=1).squeeze(0).to(torch.uint8))
            classes = torch.stack(classes)
        out = self._paster.paste(bg, faces, list(helper.inverse_affine_matrices), classes, getattr(helper, 'upscale_factor', 1), draw_box)
        return out.cpu().numpy()

    # ------------------------------------------------------------------ streamed restore + paste (round 5)
    def _stream_applies(self, helper, frames_bgr, crops, draw_box):
        """The streamed form of steps 3 + 4 needs: the engine's net (``run_clips_u8`` with ``sink``), not inside a torch.distributed
        job, the GPU cv path (``_gpu_cv_path``), the parse-mask composite with ParseNet on the engine, no face upsampler, uint8
        colour frames of one size and 512 x 512 crops -- i.e. the configuration ``_gpu_paste_applies`` admits frame by frame, decided
        once for the sequence.  ``KEEP_AMD_STREAM_PASTE=0`` keeps the per-frame path."""
        net = self.keep_net
        if os.environ.get('KEEP_AMD_STREAM_PASTE', '1') == '0' or not getattr(net, 'supports_sink', False):
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False
        if self.face_upscale_model is not None or not getattr(helper, 'use_parse', False):
            return False
        if getattr(getattr(helper, 'face_parse', None), 'engine', None) is None or not self._gpu_cv_path():
            return False
        f0 = frames_bgr[0]
        if any(not isinstance(f, np.ndarray) or f.dtype != np.uint8 or f.ndim != 3 or f.shape != f0.shape or f.shape[2] != 3
               for f in frames_bgr):
            return False
        if tuple(getattr(helper, 'face_size', (512, 512))) != (512, 512):
            return False
        return crops is None or all(tuple(c.shape) == (512, 512, 3) for c in crops)

    def _restore_and_paste_streamed(self, frames_bgr, crops, affines, faces_per_frame, max_clip_length, factor, draw_box, pbar,
                                    as_u8):
        """Steps 3 and 4 of ``process_image_sequence`` (keep_processor.py:263-304) as ONE pipeline on the GPU.

        ``run_clips_u8(sink=...)`` restores the clips in batch groups and hands each finished, range-checked group over as device
        tensors (a pool worker's share arrives in host memory, with its ParseNet class maps).  Whenever the next frames in order
        have all their faces, they are processed on a second stream: ParseNet over up to 32 faces across frames, then per frame the
        reference's arithmetic in the reference's order -- inverse affine from the helper (``get_inverse_affine``), face after face
        blended into the background (engine/paste.py) -- and the download of the finished frame (converted to the ComfyUI float RGB
        layout on the device unless ``as_u8``).  The compute stream meanwhile runs the next group's forward."""
        from ..engine import hiplib as L
        from ..engine.paste import GpuPaster
        helper, net, dev = self.face_helper, self.keep_net, torch.device(self.device)
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        if self._paster is None:
            self._paster = GpuPaster(dev)
        paster, engine = self._paster, helper.face_parse.engine
        n_frames, n_crops = len(frames_bgr), len(crops)
        spans = split_clips(n_crops, max_clip_length)
        first = [0]
        for k in faces_per_frame:
            first.append(first[-1] + k)
        clips = []
        for s, e in spans:
            part = crops[s:e]
            if all(isinstance(c, torch.Tensor) and c.is_cuda for c in part):
                clips.append(torch.stack(part))                       # warped on this GPU: restored without leaving it
            else:
                clips.append([c.cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c) for c in part])
        pool = getattr(net, 'pool', None)
        if pool is not None and hasattr(pool, 'set_parser'):          # the workers parse what they restore
            pool.set_parser(engine)
        ready, classes = [None] * n_crops, [None] * n_crops
        keep_restored = self.keep_restored_faces
        ps = getattr(self, '_paste_stream', None)
        if ps is None:
            ps = self._paste_stream = torch.cuda.Stream(device=dev)
        st = {'next': 0, 'out': None, 'aff': 0}

        def on_dev(t):
            return t if (isinstance(t, torch.Tensor) and t.is_cuda) else torch.as_tensor(t).to(dev, non_blocking=True)

        def emit(t, frame_dev):
            """finished frame t (uint8 BGR [H,W,3] on the device) -> the output tensor in host memory"""
            if st['out'] is None:
                shape = (n_frames,) + tuple(frame_dev.shape)
                try:
                    st['out'] = torch.empty(shape, dtype=torch.uint8 if as_u8 else torch.float32, pin_memory=True)
                except RuntimeError:                                  # more than the host will pin: pageable memory, slower copies
                    st['out'] = torch.empty(shape, dtype=torch.uint8 if as_u8 else torch.float32)
            if as_u8:
                st['out'][t].copy_(frame_dev, non_blocking=True)
            else:                                                      # cv2_to_comfy_image on the device (keep_bgr_u8_to_comfy: one IEEE division)
                ff = torch.empty(frame_dev.shape, dtype=torch.float32, device=dev)
                L.call('keep_bgr_u8_to_comfy', frame_dev.contiguous(), ff, frame_dev.numel() // 3)
                st['out'][t].copy_(ff, non_blocking=True)

        def advance():
            f0 = f1 = st['next']
            while f1 < n_frames and all(ready[i] is not None for i in range(first[f1], first[f1 + 1])):
                f1 += 1
            if f1 == f0:
                return
            with torch.cuda.device(dev), torch.cuda.stream(ps):
                need = [i for i in range(first[f0], first[f1]) if classes[i] is None]
                for b0 in range(0, len(need), 32):                     # ParseNet across frames, <= 32 faces per call
                    idx = need[b0:b0 + 32]
                    fb = torch.stack([on_dev(ready[i]) for i in idx])
                    x = torch.empty(fb.shape, dtype=torch.float32, device=dev)
                    L.call('keep_img2tensor', fb, x, fb.numel() // 3)     # :418-422 BGR u8 -> RGB float (x / 255 - 0.5) / 0.5
                    cl = engine.classes(x)
                    for k, i in enumerate(idx):
                        classes[i] = cl[k]
                for t in range(f0, f1):
                    bg = self._final_background(frames_bgr[t], factor)
                    k = faces_per_frame[t]
                    if k == 0:
                        emit(t, on_dev(np.ascontiguousarray(bg)))
                        continue
                    helper.affine_matrices = affines[st['aff']:st['aff'] + k]
                    helper.upscale_factor = factor
                    helper.get_inverse_affine(None)
                    st['aff'] += k
                    ids = range(first[t], first[t + 1])
                    faces = torch.stack([on_dev(ready[i]) for i in ids])
                    cls = torch.stack([on_dev(classes[i]) for i in ids])
                    emit(t, paster.paste(bg, faces, list(helper.inverse_affine_matrices), cls, factor, draw_box))
                    pbar.update(1)
                if not keep_restored:                                  # a long video must not hold every restored crop in HBM until the end
                    for i in range(first[f0], first[f1]):
                        ready[i], classes[i] = _EMITTED, None
            st['next'] = f1

        def sink(ids, crops_list, classes_list):
            for j, cid in enumerate(ids):
                s, e = spans[cid]
                for k in range(e - s):
                    ready[s + k] = crops_list[j][k]
                    if classes_list is not None:
                        classes[s + k] = classes_list[j][k]
            advance()

        max_b = None
        groups = int(os.environ.get('KEEP_AMD_STREAM_GROUPS', '0') or 0)      # >= 2: at least that many batch groups (more overlap, smaller batches)
        if groups >= 2 and hasattr(net, 'clips_per_call'):
            cap = net.clips_per_call(max_clip_length, 512, 512)
            max_b = max(1, min(cap, -(-len(clips) // groups)))
        import inspect
        try:
            params = inspect.signature(net.run_clips_u8).parameters
        except (TypeError, ValueError):
            params = {}
        takes_all = any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values())
        kw = {k: v for k, v in (('max_b', max_b), ('parse', True)) if takes_all or k in params}    # decided by capability, not by catching
        net.run_clips_u8(clips, sink=sink, **kw)                                                     # a TypeError from inside the forward
        advance()                                                      # frames without faces behind the last restored one
        ps.synchronize()
        if st['next'] != n_frames:
            raise RuntimeError(f"streamed paste-back: {n_frames - st['next']} frame(s) never became ready")
        pbar.update(n_frames)
        # the restored crops stay on the GPU only on request (tests / tools: ``keep_restored_faces``, KEEP_AMD_KEEP_RESTORED=1); by default
        # each crop is released once its frame has been composited and nothing of the sequence outlives the call in HBM
        self.last_restored_faces = _DeviceFaces(ready) if keep_restored else []
        return st['out']

    # ------------------------------------------------------------------ sequence
    def _detect_all(self, frames_bgr, only_center_face):
        """Landmarks of every frame (keep_processor.py:207-213: one ``get_face_landmarks_5`` call -- one detector forward and
        one host round trip -- per frame).  With the engine's detector (engine/retinaface.py, ``detect_batch``) the network
        runs over the video in chunks of the detector's batch axis (``KEEP_AMD_DETECT_BATCH`` frames: the host never holds more
        than one chunk of detector inputs -- a 3000-frame 1080p video would otherwise stack ~13 GB of resized frames); the
        helper's own per-frame host logic (resize rule, eye-distance filter, centre-face selection:
        face_restoration_helper.py:206-252) then runs unchanged on the stored detections of the chunk, on the SAME
        ``read_image`` result the detector input was built from, so the landmarks are what the per-frame loop produces."""
        raw = []
        helper = self.face_helper
        det = getattr(helper, 'face_detector', None)
        use_batch = hasattr(det, 'detect_batch') and getattr(helper, 'det_model', 'retinaface') != 'dlib' and len(frames_bgr) > 0
        chunk = max(1, int(getattr(getattr(det, 'engine', None), 'max_frames', 32))) if use_batch else 1
        bar = tqdm(total=len(frames_bgr), desc="Detecting face landmarks")
        starts = list(range(0, len(frames_bgr), chunk))
        # Round 6: the detector's forward of chunk k (GPU, one worker thread: the launches and the D2H waits release the GIL) runs under the
        # HOST preparation of chunk k + 1 (read_image + the INTER_AREA resize of every frame: a third of the pre-pass) -- the same calls
        # on the same data in the same order per frame, only interleaved; KEEP_AMD_DETECT_OVERLAP=0: one after the other
        pool = None
        if use_batch and len(starts) > 1 and os.environ.get('KEEP_AMD_DETECT_OVERLAP', '1') != '0':
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='keep-detect')
        try:
            prepared = self._prep_detect_chunk(frames_bgr[starts[0]:starts[0] + chunk], 640) if use_batch else None
            for k, s in enumerate(starts):
                part = frames_bgr[s:s + chunk]
                states, batched = None, None
                if use_batch:
                    states, batch = prepared
                    fut = None
                    if batch is not None:
                        fut = pool.submit(det.detect_batch, batch, 0.97) if pool is not None else None
                        if fut is None:
                            batched = det.detect_batch(batch, 0.97)
                    if k + 1 < len(starts):                                # host work of the next chunk, under this chunk's forward
                        prepared = self._prep_detect_chunk(frames_bgr[starts[k + 1]:starts[k + 1] + chunk], 640)
                    if fut is not None:
                        batched = fut.result()
                for j, frame in enumerate(part):
                    helper.clean_all()
                    if states is not None:
                        helper.input_img, helper.is_gray = states[j]       # what read_image(frame) left behind (:172-184)
                    else:
                        helper.read_image(frame)
                    if batched is None:
                        helper.get_face_landmarks_5(only_center_face=only_center_face, resize=640, eye_dist_threshold=5)
                    else:
                        helper.face_detector = _ReplayDetector(batched[j])
                        try:
                            helper.get_face_landmarks_5(only_center_face=only_center_face, resize=640, eye_dist_threshold=5)
                        finally:
                            helper.face_detector = det
                    raw.append(list(helper.all_landmarks_5))
                    bar.update(1)
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
        bar.close()
        return raw

    def _prep_detect_chunk(self, frames_bgr, resize):
        """One chunk of frames, host side only: the helper state ``read_image`` leaves behind per frame (input image, grey flag) and the
        detector inputs ``get_face_landmarks_5`` would build from it (face_restoration_helper.py:206-216: frames whose short side exceeds
        ``resize`` are scaled down with INTER_AREA), stacked -> (states, batch).  ``batch`` is None when the frames differ in size or are
        not uint8 (16-bit sources become float64 in ``read_image``: the per-frame path converts them the way the reference does)."""
        helper = self.face_helper
        states, imgs = [], []
        for frame in frames_bgr:
            helper.clean_all()
            helper.read_image(frame)
            img = helper.input_img
            states.append((img, getattr(helper, 'is_gray', False)))
            h, w = img.shape[:2]
            if resize is not None and min(h, w) > resize:
                scale = resize / min(h, w)
                own = getattr(helper, 'resize_for_detector', None)      # (a helper may bring its own resize: bench.py's cv2-free stand-in)
                img = (own(img, int(w * scale), int(h * scale)) if own is not None
                       else _resize(img, int(w * scale), int(h * scale), 'INTER_AREA' if scale < 1 else 'INTER_LINEAR'))
            imgs.append(img if isinstance(img, torch.Tensor) else np.ascontiguousarray(img))
        if any(tuple(im.shape) != tuple(imgs[0].shape) or im.dtype not in (np.uint8, torch.uint8) for im in imgs):
            return states, None
        return states, (torch.stack(imgs) if isinstance(imgs[0], torch.Tensor) else np.stack(imgs))

    def _detect_batched(self, det, frames_bgr, resize):
        """``_prep_detect_chunk`` + the detector's batched forward with the helper's 0.97 confidence threshold (:221): (states, detections)."""
        states, batch = self._prep_detect_chunk(frames_bgr, resize)
        return states, (None if batch is None else det.detect_batch(batch, 0.97))

    @torch.no_grad()
    def process_image_sequence(self, image_sequence_tensor: torch.Tensor, final_upscale_factor: float,
                               has_aligned_frames: bool, only_center_face: bool, draw_box: bool,
                               max_clip_length: int = 20):
        n_frames = image_sequence_tensor.shape[0]
        if n_frames == 0:
            return image_sequence_tensor
        # comfy_image_to_cv2 per frame (keep_processor.py:201,306 of the reference) -- on the device, chunked, under the detection pre-pass
        frames_bgr = frames_from_comfy(image_sequence_tensor, self.device)
        out = self._process_frames(frames_bgr, final_upscale_factor, has_aligned_frames, only_center_face, draw_box,
                                   max_clip_length, as_u8=False)
        if out is None:
            return None
        return out if isinstance(out, torch.Tensor) else (torch.cat([cv2_to_comfy_image(f) for f in out], dim=0) if out
                                                           else image_sequence_tensor)

    @torch.no_grad()
    def process_frames_u8(self, frames_bgr, final_upscale_factor: float = 1.0, has_aligned_frames: bool = False,
                          only_center_face: bool = True, draw_box: bool = False, max_clip_length: int = 20):
        """``process_image_sequence`` between its two ComfyUI converters: list of uint8 BGR frames -> uint8 BGR frames
        (a [N,H',W',3] tensor from the streamed path, a list of arrays otherwise).  What a video tool that already holds
        uint8 frames calls, and what bench.py times next to the float entry point."""
        return self._process_frames(list(frames_bgr), final_upscale_factor, has_aligned_frames, only_center_face, draw_box,
                                    max_clip_length, as_u8=True)

    def _process_frames(self, frames_bgr, final_upscale_factor, has_aligned_frames, only_center_face, draw_box,
                        max_clip_length, as_u8):
        n_frames = len(frames_bgr)
        pbar = ProgressBar(n_frames * 4)
        helper = self.face_helper

        # -- 1. landmarks per frame -> temporally smoothed tracks
        tracks = {}
        if not has_aligned_frames:
            raw = self._detect_all(frames_bgr, only_center_face)
            pbar.update(n_frames)
            tracks = smooth_center_face(raw) if only_center_face else smooth_tracked_faces(raw)
            pbar.update(n_frames)
        else:
            pbar.update(n_frames * 2)

        # -- 2. crops, flat and frame-major
        crops, affines, faces_per_frame = [], [], []
        stream_ok = (not has_aligned_frames) and self._stream_applies(helper, frames_bgr, None, draw_box)
        for i in tqdm(range(n_frames), desc="Cropping and aligning faces"):
            if has_aligned_frames:
                frame_crops, frame_aff = [_resize(frames_bgr[i], 512, 512, 'INTER_LINEAR')], []
            else:
                frame_crops, frame_aff = [], []
                active = [seq[i] for seq in tracks.values() if not np.isnan(seq[i]).any()]
                if active:
                    helper.clean_all()
                    helper.read_image(frames_bgr[i])
                    helper.all_landmarks_5 = active
                    self._align_warp(helper, keep_on_device=stream_ok)
                    frame_crops = list(helper.cropped_faces)
                    frame_aff = list(helper.affine_matrices)
            faces_per_frame.append(len(frame_crops))
            crops.extend(frame_crops)
            affines.extend(frame_aff)

        # -- 3 + 4 streamed: restore group g+1 while group g is parsed, pasted and downloaded (round 5)
        if crops and not has_aligned_frames and self._stream_applies(helper, frames_bgr, crops, draw_box):
            return self._restore_and_paste_streamed(frames_bgr, crops, affines, faces_per_frame, max_clip_length,
                                                    final_upscale_factor, draw_box, pbar, as_u8)

        # -- 3. restore: the hot path
        crops = [c.cpu().numpy() if isinstance(c, torch.Tensor) else c for c in crops]
        restored_faces = []
        if crops:
            restored_faces = self._restore_crops_u8(crops, max_clip_length)
            if restored_faces is None:          # non-root rank of a sharded multi-GPU run (engine/dist.py): nothing to paste here
                return None
        self.last_restored_faces = restored_faces
        pbar.update(n_frames)

        # -- 4. paste back
        out_frames = []
        face_ptr = aff_ptr = 0
        for i in tqdm(range(n_frames), desc="Pasting faces and finalizing frames"):
            bg = self._final_background(frames_bgr[i], final_upscale_factor)
            k = faces_per_frame[i]
            if has_aligned_frames and self.return_restored_aligned and k:
                # opt-in fix of reference quirk P2: an aligned sequence returns its restored faces, resized the way the
                # single-image node does (keep_processor.py:190-197), instead of the upscaled input
                face = restored_faces[face_ptr]
                face_ptr += k
                if self.face_upscale_model:
                    face = self._run_upscaler(self.face_upscale_model, face)
                side = int(512 * final_upscale_factor)
                out_frames.append(_resize(face, side, side, 'INTER_LANCZOS4'))
                continue
            if k == 0 or has_aligned_frames:            # aligned: restored faces unused (reference quirk P2)
                out_frames.append(bg)
                continue
            helper.restored_faces = [f.astype('uint8') for f in restored_faces[face_ptr:face_ptr + k]]
            helper.affine_matrices = affines[aff_ptr:aff_ptr + k]
            helper.upscale_factor = final_upscale_factor
            helper.get_inverse_affine(None)
            out_frames.append(self._paste(helper, bg, draw_box))
            face_ptr += k
            aff_ptr += k
            pbar.update(1)

        return out_frames
