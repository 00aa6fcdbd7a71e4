"""CPU suite: host-side logic of the drop-in surface -- converters (P4), clip chunking / ordering (P1-P3),
weight ingestion (P5), node surface (8b), C-ABI exports."""
import ctypes
import os
import re
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT


# ------------------------------------------------------------------ stub ComfyUI runtime for the node surface
def _install_comfy_stub():
    if 'comfy' in sys.modules:
        return
    comfy = types.ModuleType('comfy')
    mm = types.ModuleType('comfy.model_management')
    mm.get_torch_device = lambda: torch.device('cpu')
    mm.unet_offload_device = lambda: torch.device('cpu')
    mm.soft_empty_cache = lambda: None
    cu = types.ModuleType('comfy.utils')

    class ProgressBar:
        last = None

        def __init__(self, total):
            self.total, self.current = total, 0
            ProgressBar.last = self

        def update(self, n):
            self.current += n

    cu.ProgressBar = ProgressBar
    cu.tiled_scale = None
    comfy.model_management, comfy.utils = mm, cu
    fp = types.ModuleType('folder_paths')
    fp.models_dir = '/nonexistent/models'
    sys.modules.update({'comfy': comfy, 'comfy.model_management': mm, 'comfy.utils': cu, 'folder_paths': fp})


_install_comfy_stub()
import importlib  # noqa: E402

nodes = importlib.import_module('comfyui_keep_amd.nodes')
from comfyui_keep_amd.modules import utils as U  # noqa: E402
from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack, convert_legacy_keys, select_params  # noqa: E402
from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor, split_clips  # noqa: E402
from comfyui_keep_amd.engine import synth  # noqa: E402


def test_node_surface():
    assert set(nodes.NODE_CLASS_MAPPINGS) == {'KEEP_ModelLoader', 'KEEP_FaceUpscaleImage', 'KEEP_ProcessImageSequence'}
    assert nodes.NODE_DISPLAY_NAME_MAPPINGS == {
        'KEEP_ModelLoader': 'Load KEEP Models', 'KEEP_FaceUpscaleImage': 'KEEP Single Image',
        'KEEP_ProcessImageSequence': 'KEEP Image Sequence'}
    seq = nodes.KEEP_ProcessImageSequenceNode
    req = seq.INPUT_TYPES()['required']
    assert list(req) == ['images', 'keep_model', 'final_upscale_factor', 'has_aligned_frames', 'only_center_face',
                         'draw_bounding_box', 'max_clip_length']
    assert req['max_clip_length'][1]['default'] == 20 and req['max_clip_length'][1]['max'] == 100
    assert req['final_upscale_factor'][1] == {**req['final_upscale_factor'][1], 'default': 1.0, 'min': 0.5, 'max': 4.0, 'step': 0.1}
    assert seq.RETURN_TYPES == ('IMAGE',) and seq.FUNCTION == 'process_sequence' and seq.CATEGORY == 'ComfyUI-KEEP'
    ld = nodes.KEEP_ModelLoaderNode
    assert ld.RETURN_TYPES == ('KEEP_MODEL_PACK',) and ld.FUNCTION == 'load_model_pack'
    it = ld.INPUT_TYPES()
    assert it['required']['model'][0] == ['KEEP', 'Asian']
    assert it['required']['detection_model'][0] == ['retinaface_resnet50', 'retinaface_mobile0.25', 'YOLOv5l', 'YOLOv5n']
    assert set(it['optional']) == {'bg_upscale_model', 'face_upscale_model'}


def test_processing_nodes_never_raise():
    img = torch.zeros(1, 8, 8, 3)
    assert nodes.KEEP_FaceUpscaleImageNode().upscale_face_image(img, "not a pack", 1.0, True, True, False) == (None,)

    class Boom:
        def __call__(self, *a, **k):
            raise RuntimeError("boom")

        def to(self, d):
            return self

    class Helper:
        pass

    pack = KEEPModelPack(Boom(), Helper(), None, None, 'KEEP')
    out = nodes.KEEP_ProcessImageSequenceNode().process_sequence(torch.zeros(2, 512, 512, 3), pack, 1.0, True, True, False, 20)
    assert out == (None,)


def test_config_table():
    a = U.KEEP_MODEL_CONFIGS['KEEP']['architecture']
    assert a['cft_list'] == ['16', '32', '64'] and a['temp_reg_list'] == ['32'] and a['kalman_attn_head_dim'] == 48
    assert U.KEEP_MODEL_CONFIGS['Asian']['architecture']['cft_list'] == ['32', '64', '128', '256']
    assert a['fix_modules'] == ['quantize', 'generator'] and a['latent_size'] == 256 and a['mask_ratio'] == 0.


def test_converters_edge_values():
    # comfy_image_to_cv2 truncates; channel order flips
    img = torch.tensor([[[[0.999, 0.5, 0.0]]]])
    assert U.comfy_image_to_cv2(img).tolist() == [[[0, 127, 254]]]
    # tensor2img: clamp, (x+1)/2*255, round half to even, RGB->BGR
    vals = torch.tensor([-1.2, -1.0, -0.5 / 255, 0.0, 1.0 / 255, 1.0, 1.3, 0.00392156862])
    t = vals.view(1, 1, -1).repeat(3, 1, 1)
    got = U.net_output_to_bgr_u8(t)[0, :, 0]
    exp = np.round((np.clip(vals.numpy(), -1, 1) + 1) / 2 * 255.0).astype(np.uint8)
    assert got.tolist() == exp.tolist()
    assert got[3] == 128 and got[0] == 0 and got[6] == 255            # 127.5 -> 128 (half to even)
    # crops -> net input: float32(u8/255.) then (x-0.5)/0.5, BGR->RGB
    crop = synth.ramp_image(4, 4)
    x = U.crops_to_net_input([crop])
    ref = ((crop / 255.).astype(np.float32)[..., ::-1].transpose(2, 0, 1) - np.float32(0.5)) / np.float32(0.5)
    assert np.array_equal(x[0].numpy(), ref)
    # round trip of exact grid values
    assert np.array_equal(U.net_output_to_bgr_u8(x[0]), crop)


@pytest.mark.parametrize("n,l", [(1, 20), (2, 20), (19, 20), (20, 20), (21, 20), (41, 20), (5, 1)])
def test_chunk_boundaries(n, l):
    chunks = split_clips(n, l)
    assert chunks[0][0] == 0 and chunks[-1][1] == n
    assert all(e - s <= l and e > s for s, e in chunks)
    assert all(chunks[i][1] == chunks[i + 1][0] for i in range(len(chunks) - 1))
    assert len(chunks) == -(-n // l)


class _RecordingNet:
    """keep_net stand-in: output = input + clip_index, records the clip lengths it was given."""

    def __init__(self):
        self.calls = []

    def __call__(self, x, need_upscale=True):
        assert need_upscale is False
        self.calls.append(x.shape[1])
        return x + float(len(self.calls))

    def to(self, d):
        return self


class _Helper:
    def clean_all(self):
        pass


def _pack(net):
    return KEEPModelPack(net, _Helper(), None, None, 'KEEP')


def test_sequence_clip_loop_aligned():
    net = _RecordingNet()
    proc = KEEPFaceProcessor(_pack(net))
    frames = torch.rand(41, 512, 512, 3)
    out = proc.process_image_sequence(frames, 1.0, True, True, False, max_clip_length=20)
    assert net.calls == [20, 20, 2]                    # 41 = 20 + 20 + 1, the singleton is duplicated to T=2
    assert len(proc.last_restored_faces) == 41
    # reference quirk P2: aligned sequences return the (resized) input, restored faces are discarded
    assert out.shape == (41, 512, 512, 3)
    assert torch.equal(out, (frames * 255).to(torch.uint8).float() / 255.0)
    from comfy.utils import ProgressBar
    assert ProgressBar.last.current == 41 * 3          # no paste ticks on the aligned path (Appendix A.2)


def test_sequence_aligned_fix_flag_returns_restored_faces():
    """Opt-in fix of reference quirk P2: with return_restored_aligned the aligned-sequence node returns the restored
    faces (here 255 - crop through the uint8 entry point) instead of its input; default stays bug-compatible."""
    net = _U8Net(True)
    proc = KEEPFaceProcessor(_pack(net))
    frames = torch.rand(3, 512, 512, 3)
    u8 = (frames * 255).to(torch.uint8)
    out_default = proc.process_image_sequence(frames, 1.0, True, True, False, max_clip_length=2)
    assert torch.equal(out_default, u8.float() / 255.0)
    proc.return_restored_aligned = True
    out_fixed = proc.process_image_sequence(frames, 1.0, True, True, False, max_clip_length=2)
    assert out_fixed.shape == (3, 512, 512, 3)
    assert torch.equal(out_fixed, (255 - u8).float() / 255.0)


def test_single_image_aligned_duplicates_to_T2():
    net = _RecordingNet()
    proc = KEEPFaceProcessor(_pack(net))
    img = synth.ramp_image()
    out = proc.process_image(img, 1.0, True, True, False)
    assert net.calls == [2] and out.shape == (512, 512, 3) and out.dtype == np.uint8


class _U8Net(_RecordingNet):
    """engine stand-in offering the device-side uint8 entry point: output clip = 255 - input, records clip lengths."""

    def __init__(self, single_frame):
        super().__init__()
        self.supports_single_frame = single_frame

    def run_clips_u8(self, clips, max_b=4):
        # contract of KeepNet.run_clips_u8: a clip is a uint8 [T,H,W,3] tensor or a list of T uint8 [H,W,3] crops
        clips = [torch.from_numpy(np.stack(c)) if isinstance(c, list) else c for c in clips]
        for c in clips:
            assert c.dtype == torch.uint8 and c.dim() == 4 and c.shape[1:] == (512, 512, 3)
            self.calls.append(c.shape[0])
        return [255 - c for c in clips]


@pytest.mark.parametrize("single_frame", [True, False])
def test_uint8_entry_point_chunking_and_lone_crops(single_frame):
    """SURVEY 8f-1 host logic: crops are handed over as uint8 clips chunked like keep_processor.py:263-270; a lone crop goes
    as T=1 when the engine supports it, else as the reference's T=2 duplicate with frame 0 kept -- same restored faces."""
    net = _U8Net(single_frame)
    proc = KEEPFaceProcessor(_pack(net))
    base = synth.ramp_image()
    crops = [np.ascontiguousarray(np.roll(base, k, axis=0)) for k in range(5)]
    faces = proc._restore_crops_u8(crops, 2)
    assert net.calls == ([2, 2, 1] if single_frame else [2, 2, 2])
    assert len(faces) == 5 and all(f.dtype == np.uint8 and f.shape == (512, 512, 3) for f in faces)
    assert all(np.array_equal(f, 255 - c) for f, c in zip(faces, crops))
    out = proc.process_image(base, 1.0, True, True, False)           # single-image node, aligned input
    assert net.calls[-1] == (1 if single_frame else 2) and np.array_equal(out, 255 - base)


def test_legacy_key_conversion_and_param_selection():
    sd = {'cross_fuse.16.norm1.weight': 1, 'fuse_convs_dict.32.scale.0.bias': 2, 'encoder.blocks.0.weight': 3}
    out = convert_legacy_keys(sd)
    assert set(out) == {'cfa.16.norm1.weight', 'cft.32.scale.0.bias', 'encoder.blocks.0.weight'}
    assert select_params({'params_ema': 'a', 'params': 'b'}) == 'a'
    assert select_params({'params': 'b'}) == 'b'
    plain = {'x': 1}
    assert select_params(plain) is plain


def test_strict_load_rejects_bad_state_dict(synth_weights):
    from comfyui_keep_amd.engine.net import KeepNet
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
    net = KeepNet(**DEFAULT_ARCH)
    bad = dict(synth_weights)
    bad.pop('position_emb')
    with pytest.raises(RuntimeError, match='Missing key'):
        net.load_state_dict(bad, strict=True)
    bad = dict(synth_weights, extra=torch.zeros(1))
    with pytest.raises(RuntimeError, match='Unexpected key'):
        net.load_state_dict(bad, strict=True)


def test_gmflow_conv1_as_a_4x4_convolution_on_the_space_to_depth_image():
    """engine/weights.py:s2d_weights_7x7 + the layout keep_rgb_s2d writes: the 7x7 stride-2 pad-3 convolution of GMFlow's encoder
    (GM/backbone.py:69) equals a 4x4 stride-1 convolution (pad 2 top / left, 1 bottom / right) on the 2x2 space-to-depth image with
    channel (dy*2 + dx)*3 + c -- float64, every output."""
    from comfyui_keep_amd.engine.weights import s2d_weights_7x7
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 20, 28, generator=g, dtype=torch.float64)
    w = torch.randn(8, 3, 7, 7, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, w, stride=2, padding=3)
    ws = s2d_weights_7x7(w.permute(0, 2, 3, 1).contiguous())                       # packed [Cout,7,7,3] -> [Cout,4,4,16]
    assert ws.shape == (8, 4, 4, 16) and float(ws[:, 0, :, 0:6].abs().max()) == 0.0 and float(ws[..., 12:].abs().max()) == 0.0
    N, C, H, W = x.shape
    s2d = x.view(N, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(N, 12, H // 2, W // 2)      # channel (dy*2+dx)*3 + c
    s2d = torch.nn.functional.pad(s2d, (2, 1, 2, 1))
    got = torch.nn.functional.conv2d(s2d, ws[..., :12].permute(0, 3, 1, 2))
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-12


def test_packed_blob_layouts(synth_weights):
    from comfyui_keep_amd.engine.weights import logical_tensors, pack_blob, views
    lt = logical_tensors(synth_weights)
    w = synth_weights['encoder.blocks.1.conv1.weight']
    assert torch.equal(lt['encoder.blocks.1.conv1.weight'], w.permute(0, 2, 3, 1))
    assert lt['encoder.blocks.17.qkv.weight'].shape == (1536, 512)
    assert torch.equal(lt['encoder.blocks.17.qkv.weight'][512:1024], synth_weights['encoder.blocks.17.k.weight'].view(512, 512))
    assert lt['cft.32.ss0.weight'].shape == (512, 3, 3, 256)
    assert lt['cfa.16.attn.to_kv.weight'].shape == (2048, 512)
    assert lt['flownet.model.transformer.layers.0.self_attn.qkv.weight'].shape == (384, 128)
    assert 'encoder.blocks.17.q.weight' not in lt
    blob, index = pack_blob(lt)
    v = views(torch.from_numpy(blob), index)
    assert all(off % 64 == 0 for off, _ in index.values())
    assert torch.equal(v['position_emb'], synth_weights['position_emb'])
    # the one ragged 3x3 convolution (GMFlow upsampler.0, Cin = 2 + 128) carries zero input columns up to 144
    up, src = lt['flownet.model.upsampler.0.weight'], synth_weights['flownet.model.upsampler.0.weight']
    assert up.shape == (256, 3, 3, 144) and torch.equal(up[..., :130], src.permute(0, 2, 3, 1)) and not up[..., 130:].any()
    n_in = sum(t.numel() for t in synth_weights.values())
    # ... and GMFlow's 7x7 stride-2 first convolution a derived [64,4,4,16] twin for the space-to-depth form (keep_rgb_s2d)
    assert lt['flownet.model.backbone.conv1.weight_s2d'].shape == (64, 4, 4, 16)
    # ... and (round 6) every CFT block's encode_enc first convolution and 1x1 shortcut once more as their encoder / decoder input-channel
    # halves (engine/net.py:_cft_enc_part): the same numbers, sliced
    for sz, C in (('16', 512), ('32', 256), ('64', 256)):
        q = f'cft.{sz}.encode_enc'
        full, sc = lt[f'{q}.conv1.weight'], lt[f'{q}.conv_out.weight']
        assert full.shape == (C, 3, 3, 2 * C) and sc.shape == (C, 2 * C)
        assert torch.equal(torch.cat([lt[f'{q}.conv1.weight_enc'], lt[f'{q}.conv1.weight_dec']], -1), full)
        assert torch.equal(torch.cat([lt[f'{q}.conv_out.weight_enc'], lt[f'{q}.conv_out.weight_dec']], -1), sc)
    halves = sum(C * 9 * 2 * C + C * 2 * C for C in (512, 256, 256))
    assert sum(t.numel() for t in lt.values()) == n_in + 256 * 9 * 14 + 64 * 4 * 4 * 16 + halves


def test_face_tracking_restatement():
    from comfyui_keep_amd.modules.face_tracks import interpolate_sequence, smooth_center_face, track_faces
    s = np.array([1.0, np.nan, 3.0, np.nan, np.nan, 6.0])
    assert np.allclose(interpolate_sequence(s), [1, 2, 3, 4, 5, 6])
    lm = lambda cx: np.tile(np.array([[cx, 10.0]]), (5, 1))  # noqa: E731
    tracks = track_faces([[lm(10), lm(300)], [lm(305), lm(12)], [], [lm(14)]])
    assert len(tracks) == 3                                    # the face re-appearing after a gap starts a new track
    assert np.allclose(tracks[0][1], lm(12)) and np.allclose(tracks[1][1], lm(305))
    assert np.isnan(tracks[0][2]).all() and np.isnan(tracks[0][3]).all()
    sm = smooth_center_face([[lm(10)], [], [lm(30)]])
    assert sm[0].shape == (3, 5, 2) and not np.isnan(sm[0]).any()


def test_model_pack_moves_a_yolo_detector_that_has_no_to():
    """the reference's YoloDetector is a plain class without .to(); load_device() / offload() must move its network
    (`.detector`) and retarget `.device` instead of raising AttributeError."""
    class _Net:
        def __init__(self):
            self.where = None

        def to(self, d):
            self.where = d
            return self

    class _Yolo:
        def __init__(self):
            self.detector, self.device = _Net(), 'cpu'

    helper = _Helper()
    helper.face_detector, helper.face_parse = _Yolo(), _Net()
    pack = KEEPModelPack(None, helper, None, None, 'KEEP')
    pack.load_device()
    assert helper.face_detector.detector.where == pack.device and helper.face_detector.device == pack.device
    assert helper.face_parse.where == pack.device and helper.device == pack.device
    pack.offload()
    assert helper.face_detector.detector.where == pack.offload_device and helper.face_detector.device == pack.offload_device


def test_c_abi_exports_every_declared_symbol():
    from comfyui_keep_amd.engine import hiplib
    header = open(os.path.join(ROOT, 'include', 'keep_hip.h')).read()
    declared = set(re.findall(r'\b(keep_[a-z0-9_]+)\s*\(', header))
    assert declared == set(hiplib.EXPORTED_SYMBOLS)
    assert os.path.exists(hiplib.LIB_PATH), "run `python __graft_entry__.py` (build) first"
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.keep_abi_version.restype = ctypes.c_int32
    assert lib.keep_abi_version() == hiplib.ABI_VERSION
    hiplib.load(check_device=False)                             # binds every signature
    # argument validation is host-side C: callable without a GPU
    lib.keep_last_error.restype = ctypes.c_char_p
    lib.keep_conv2d.restype = ctypes.c_int32
    assert lib.keep_conv2d(None, None) == -1 and b'null args' in lib.keep_last_error()


def _header_struct_fields(header, name):
    """Field names of `typedef struct { ... } name;` in declaration order (comments stripped, `a, b, c;` lists expanded)."""
    end = re.search(r'\}\s*' + name + r'\s*;', header).start()
    body = header[header.rindex('typedef struct {', 0, end) + len('typedef struct {'):end]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(',')
        m = re.search(r'(\w+)\s*(\[\d+\])?$', first.strip())
        names.append(m.group(1))
        names += [r.strip().lstrip('*').strip() for r in rest]
    return names


def test_abi_struct_layouts_match_header_library_and_doc():
    """The three descriptions of the argument structs -- include/keep_hip.h, the ctypes binding, the built library -- and
    the copy printed in INTEGRATION.md agree: field names and order (header vs binding), byte size (binding vs
    keep_sizeof_*_args() of the built .so), documented listing (generated from the binding, tools/gen_abi_doc.py)."""
    from comfyui_keep_amd.engine import hiplib
    header = open(os.path.join(ROOT, 'include', 'keep_hip.h')).read()
    assert int(re.search(r'#define KEEP_ABI_VERSION (\d+)', header).group(1)) == hiplib.ABI_VERSION
    rename = {'inp': 'in'}
    for cname, cls in (('keep_conv2d_args', hiplib.ConvArgs), ('keep_conv2d_plan_out', hiplib.ConvPlanOut),
                       ('keep_attention_args', hiplib.AttnArgs)):
        assert [rename.get(n, n) for n, _ in cls._fields_] == _header_struct_fields(header, cname), cname
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    lib.keep_sizeof_conv2d_args.restype = lib.keep_sizeof_attention_args.restype = ctypes.c_int32
    assert lib.keep_sizeof_conv2d_args() == ctypes.sizeof(hiplib.ConvArgs)
    assert lib.keep_sizeof_attention_args() == ctypes.sizeof(hiplib.AttnArgs)
    assert hiplib.ConvArgs._fields_[0][0] == 'struct_size' and hiplib.AttnArgs._fields_[0][0] == 'struct_size'
    # the v12 minimum sizes the header promises to keep accepting are not larger than today's structs
    for macro, cls in (('KEEP_CONV2D_ARGS_V12_SIZE', hiplib.ConvArgs), ('KEEP_ATTENTION_ARGS_V12_SIZE', hiplib.AttnArgs)):
        assert int(re.search(rf'#define {macro} (\d+)', header).group(1)) <= ctypes.sizeof(cls)
    # a struct_size the library does not know is refused before anything is read
    lib.keep_last_error.restype = ctypes.c_char_p
    lib.keep_conv2d_plan.restype = ctypes.c_int32
    a, out = hiplib.ConvArgs(), hiplib.ConvPlanOut()
    for bad in (0, 128, ctypes.sizeof(hiplib.ConvArgs) + 8):
        a.struct_size = bad
        assert lib.keep_conv2d_plan(ctypes.byref(a), ctypes.byref(out)) == -1 and b'struct_size' in lib.keep_last_error()
    # the plan query itself is host-side C: a well-formed struct is answered without a GPU
    a = hiplib.ConvArgs(struct_size=ctypes.sizeof(hiplib.ConvArgs), N=2, H=64, W=64, Cin=128, Cout=128, KH=3, KW=3, stride=1,
                        pad_t=1, pad_l=1, Ho=64, Wo=64, in_ld=128, out_ld=128, mma=hiplib.MMA_F32)
    assert lib.keep_conv2d_plan(ctypes.byref(a), ctypes.byref(out)) == 0, lib.keep_last_error()
    assert out.kernel.decode().startswith('conv3x3_halo_f32_kernel')


def test_library_reads_no_environment_variable_and_flags_steer_the_plan(monkeypatch):
    """the product libkeep_hip.so has no getenv (dev switches exist only under -DKEEP_DEV_KNOBS); what used to be
    KEEP_NO_* / KEEP_PLAN_REF_IMAGES / KEEP_GATHER_SMALL_M travels in keep_conv2d_args.flags / .plan_ref_images."""
    import subprocess
    from comfyui_keep_amd.engine import hiplib
    csrc = os.path.join(ROOT, 'comfyui-keep_amd', 'csrc')
    n_getenv = sum(open(os.path.join(csrc, f)).read().count('getenv') for f in os.listdir(csrc) if f.endswith(('.hip', '.h')))
    assert n_getenv <= 2, n_getenv          # the KEEP_DEV_ENV macro's definition + its comment, both behind #ifdef KEEP_DEV_KNOBS
    syms = subprocess.run(['nm', '-D', '--undefined-only', hiplib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'getenv' not in syms, "the default build must not import getenv"
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    lib.keep_conv2d_plan.restype = ctypes.c_int32
    lib.keep_last_error.restype = ctypes.c_char_p

    def plan(**kw):
        base = dict(struct_size=ctypes.sizeof(hiplib.ConvArgs), N=1, H=64, W=64, Cin=128, Cout=128, KH=3, KW=3, stride=1, pad_t=1,
                    pad_l=1, Ho=64, Wo=64, in_ld=128, out_ld=128, mma=hiplib.MMA_F32)
        base.update(kw)
        out = hiplib.ConvPlanOut()
        assert lib.keep_conv2d_plan(ctypes.byref(hiplib.ConvArgs(**base)), ctypes.byref(out)) == 0, lib.keep_last_error()
        return out.kernel.decode(), out.split_k

    ref = plan()
    assert ref[0].startswith('conv3x3_halo_f32_kernel')
    for k in ('KEEP_NO_HALO_F32', 'KEEP_NO_HALO_X3', 'KEEP_NO_COUT4', 'KEEP_NO_C3', 'KEEP_NO_PLAIN', 'KEEP_X3_NO_STREAM'):
        monkeypatch.setenv(k, '1')
    monkeypatch.setenv('KEEP_PLAN_REF_IMAGES', '1')
    monkeypatch.setenv('KEEP_GATHER_SMALL_M', '1')
    assert plan() == ref                                                          # the environment is not looked at
    assert not plan(flags=hiplib.CONV_NO_HALO_F32)[0].startswith('conv3x3_halo_f32_kernel')      # ... the flag is
    # the reference batch of the plans: 16 images of 16x16 need no deeper split than 1 image's plan under reference 1
    g16 = dict(H=16, W=16, Ho=16, Wo=16, Cin=512, Cout=512, in_ld=512, out_ld=512)
    assert plan(plan_ref_images=1, **g16)[1] > plan(**g16)[1] and plan(plan_ref_images=16, **g16) == plan(**g16)
    small = dict(H=2304, W=1, Ho=2304, Wo=1, KH=1, KW=1, pad_t=0, pad_l=0, Cin=128, Cout=192, in_ld=128, out_ld=192)
    assert '2, 2, 2, 2' in plan(**small)[0] and '2, 2, 1, 1' in plan(flags=hiplib.CONV_SMALL_TILES, **small)[0]


def test_integration_doc_matches_the_binding():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_abi_doc
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    a, b = doc.index(gen_abi_doc.BEGIN), doc.index(gen_abi_doc.END) + len(gen_abi_doc.END)
    assert doc[a:b] == gen_abi_doc.block(), "INTEGRATION.md section 3 is stale: run python tools/gen_abi_doc.py"


def test_plan_is_batch_invariant_in_the_parity_policies():
    """keep_conv2d_plan (host-side C, no GPU needed): under the exact-f32 policy the split-K factor, the kernel and the
    statistics partition of a layer are the same for 1, 3, 16 and 320 images -- a clip's sums never depend on its
    batch-mates."""
    from comfyui_keep_amd.engine import hiplib
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    lib.keep_conv2d_plan.restype = ctypes.c_int32
    lib.keep_last_error.restype = ctypes.c_char_p
    geoms = [(16, 16, 512, 512, 3), (32, 32, 256, 256, 3), (64, 64, 256, 128, 3), (256, 256, 128, 128, 3), (256, 1, 512, 1024, 1),
             (4096, 1, 128, 384, 1), (16, 16, 512, 1536, 1)]
    for H, W, Cin, Cout, k in geoms:
        seen = set()
        for N in (1, 3, 16, 320):
            a = hiplib.ConvArgs(struct_size=ctypes.sizeof(hiplib.ConvArgs), N=N, H=H, W=W, Cin=Cin, Cout=Cout, KH=k, KW=k,
                                stride=1, pad_t=k // 2, pad_l=k // 2, Ho=H, Wo=W, in_ld=Cin, out_ld=Cout, mma=hiplib.MMA_F32)
            out = hiplib.ConvPlanOut()
            assert lib.keep_conv2d_plan(ctypes.byref(a), ctypes.byref(out)) == 0, lib.keep_last_error()
            seen.add((out.split_k, out.kernel, out.stats_rows, out.stats_P))
        assert len(seen) == 1, (H, W, Cin, Cout, k, seen)


def test_plan_layernorm_epilogue_rules():
    """keep_conv2d_plan (host-side C): ln_gamma is accepted only where the LayerNorm epilogue exists -- the x3 GEMM form with
    128 output channels and whole 128-row tiles -- and refused with KEEP_EUNSUP (-2) everywhere else, never silently dropped."""
    from comfyui_keep_amd.engine import hiplib
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    lib.keep_conv2d_plan.restype = ctypes.c_int32
    lib.keep_last_error.restype = ctypes.c_char_p
    buf = (ctypes.c_float * 4096)()
    ptr = (ctypes.addressof(buf) + 63) // 64 * 64

    def plan(**kw):
        base = dict(struct_size=ctypes.sizeof(hiplib.ConvArgs), N=2, H=4096, W=1, Cin=128, Cout=128, KH=1, KW=1, stride=1, pad_t=0,
                    pad_l=0, Ho=4096, Wo=1, in_ld=128, out_ld=128, mma=hiplib.MMA_X3, inp=ptr, out=ptr, weight=ptr, weight_x3=ptr,
                    x3_acc_scale=1.0, ln_gamma=ptr, ln_beta=ptr, ln_eps=1e-5)
        base.update(kw)
        out = hiplib.ConvPlanOut()
        rc = lib.keep_conv2d_plan(ctypes.byref(hiplib.ConvArgs(**base)), ctypes.byref(out))
        return rc, out.kernel.decode(), out.split_k

    rc, kernel, sk = plan()
    assert rc == 0 and 'LayerNorm' in kernel and '<4, 1, 1, 4' in kernel and sk == 1, (rc, kernel, lib.keep_last_error())
    assert plan(Cin=1024, in_ld=1024)[0] == 0
    assert plan(ln_gamma=None, ln_beta=None)[1].startswith('conv_x3_kernel<2, 2, 2, 2')         # unchanged without it
    for bad in (dict(Cout=256, out_ld=256), dict(H=4000, Ho=4000), dict(epi_act=hiplib.ACT_GELU), dict(mma=hiplib.MMA_F32),
                dict(KH=3, KW=3, pad_t=1, pad_l=1, H=64, W=64, Ho=64, Wo=64), dict(split_k=2), dict(ln_beta=None)):
        rc, kernel, _ = plan(**bad)
        assert rc == -2 and b'ln_gamma' in lib.keep_last_error(), (bad, rc, kernel)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'comfyui-keep_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'keep_oracle' not in src and 'ref_import' not in src and 'import oracle' not in src, f


def test_paste_hook_defers_to_the_helper():
    """the GPU paste is opt-in and only takes the configuration it restates; everything else is the helper's
    own paste_faces_to_input_image with the reference's arguments."""
    calls = []

    class Hp(_Helper):
        use_parse, is_gray, upscale_factor, face_size = True, False, 1, (512, 512)
        face_parse = object()

        def paste_faces_to_input_image(self, upsample_img=None, draw_box=False, face_upsampler=None):
            calls.append((draw_box, face_upsampler))
            return upsample_img

    h = Hp()
    proc = KEEPFaceProcessor(KEEPModelPack(_RecordingNet(), h, None, None, 'KEEP'))
    bg = np.zeros((64, 64, 3), np.uint8)
    h.input_img, h.restored_faces, h.inverse_affine_matrices = bg, [np.zeros((512, 512, 3), np.uint8)], [np.eye(2, 3)]
    # unset environment = 'auto': the first paste self-checks the HIP path against cv2; without cv2 (this image) / without a GPU
    # it settles on the helper's own path
    assert proc.gpu_paste is None and proc._paste(h, bg, False) is bg and calls == [(False, None)]
    assert proc.gpu_paste is False
    proc.gpu_paste = True
    assert proc._gpu_paste_applies(h, bg, False)
    assert proc._gpu_paste_applies(h, bg, True)                                      # draw_box: on the device too (keep_draw_box)
    assert not proc._gpu_paste_applies(h, bg.astype(np.float32), False)              # not uint8
    assert not proc._gpu_paste_applies(h, np.zeros((32, 32, 3), np.uint8), False)    # background still to be resized
    h.use_parse = False
    # the erosion-mask path (face_restoration_helper.py:386-415) is on the device too, but it is not part of the cv2 self-check:
    # only an explicit KEEP_AMD_GPU_PASTE=1 turns it on
    assert not proc._gpu_paste_applies(h, bg, False)
    proc._gpu_paste_forced = True
    assert proc._gpu_paste_applies(h, bg, False)
    h.use_parse, proc.face_upscale_model = True, object()
    assert not proc._gpu_paste_applies(h, bg, False)


def test_batched_detection_prepass_equals_the_per_frame_loop():
    """SURVEY 8f-4 host logic: with a detector that offers ``detect_batch`` the processor runs ONE batched detection over the
    video and lets the helper post-process each frame from the stored result; landmarks must equal the per-frame loop's
    (keep_processor.py:207-213), the helper's own ``get_face_landmarks_5`` is called once per frame either way."""
    class Det:
        def __init__(self):
            self.single_calls, self.batch_calls = 0, 0

        def _res(self, img):
            k = float(img[0, 0, 0])                      # result depends on the frame content only
            return np.array([[k, 1, k + 50, 60, 0.99] + [k + 10, 20, k + 30, 20, k + 20, 30, k + 12, 40, k + 28, 40]], np.float32)

        def detect_faces(self, img, thr=0.8):
            self.single_calls += 1
            return self._res(img)

    class BatchDet(Det):
        def detect_batch(self, frames, thr=0.8):
            self.batch_calls += 1
            assert frames.ndim == 4 and thr == 0.97
            return [self._res(f) for f in frames]

    class Helper:
        det_model = 'retinaface_resnet50'

        def __init__(self, det):
            self.face_detector = det
            self.calls = 0

        def clean_all(self):
            self.all_landmarks_5, self.input_img = [], None

        def read_image(self, img):
            self.input_img = img

        def get_face_landmarks_5(self, only_center_face=False, resize=640, eye_dist_threshold=None):
            self.calls += 1
            b = self.face_detector.detect_faces(self.input_img, 0.97)
            self.all_landmarks_5 = [b[i, 5:].reshape(5, 2) for i in range(b.shape[0])]
            return len(self.all_landmarks_5)

    frames = [np.full((64, 80, 3), 7 * i, np.uint8) for i in range(5)]
    out = {}
    for name, det in (('loop', Det()), ('batch', BatchDet())):
        helper = Helper(det)
        proc = KEEPFaceProcessor(KEEPModelPack(_RecordingNet(), helper, None, None, 'KEEP'))
        out[name] = proc._detect_all(frames, True)
        assert helper.calls == 5 and helper.face_detector is det
        assert (det.single_calls, det.batch_calls) == ((5, 0) if name == 'loop' else (0, 1))
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(out['loop'], out['batch']))


def test_detection_prepass_overlaps_chunks_without_changing_results(monkeypatch):
    """Round 6: with several chunks of detector frames the forward of chunk k runs on a worker thread under the host preparation of
    chunk k + 1 (KEEP_AMD_DETECT_OVERLAP, default on).  Same landmarks as the one-after-the-other order and as the per-frame loop, every
    chunk prepared exactly once and in order, the detector called once per chunk from ONE worker thread; an exception inside the
    detector's forward surfaces in the caller."""
    import threading
    import types
    log = []

    class Det:
        def __init__(self, fail_at=None):
            self.engine = types.SimpleNamespace(max_frames=4)         # chunks of four frames
            self.threads, self.fail_at, self.n = set(), fail_at, 0

        def detect_batch(self, frames, thr=0.8):
            self.threads.add(threading.get_ident())
            log.append(('forward', int(frames[0, 0, 0, 0])))
            self.n += 1
            if self.fail_at == self.n:
                raise RuntimeError('detector down')
            return [np.array([[float(f[0, 0, 0]), 1, float(f[0, 0, 0]) + 50, 60, 0.99] + list(np.arange(10, dtype=np.float32) + float(f[0, 0, 0]))], np.float32)
                    for f in frames]

    class Helper:
        det_model = 'retinaface_resnet50'

        def __init__(self, det):
            self.face_detector = det

        def clean_all(self):
            self.all_landmarks_5, self.input_img = [], None

        def read_image(self, img):
            self.input_img = img
            log.append(('read', int(img[0, 0, 0])))

        def get_face_landmarks_5(self, only_center_face=False, resize=640, eye_dist_threshold=None):
            b = self.face_detector.detect_faces(self.input_img, 0.97)
            self.all_landmarks_5 = [b[i, 5:].reshape(5, 2) for i in range(b.shape[0])]
            return len(self.all_landmarks_5)

    frames = [np.full((32, 40, 3), i, np.uint8) for i in range(10)]
    outs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('KEEP_AMD_DETECT_OVERLAP', mode)
        del log[:]
        det = Det()
        proc = KEEPFaceProcessor(KEEPModelPack(_RecordingNet(), Helper(det), None, None, 'KEEP'))
        outs[mode] = proc._detect_all(frames, True)
        assert [v for k, v in log if k == 'read'] == list(range(10))               # every frame prepared once, in order
        assert [v for k, v in log if k == 'forward'] == [0, 4, 8]                  # one forward per chunk, in order
        assert len(det.threads) == 1 and (threading.get_ident() in det.threads) == (mode == '0')
        if mode == '1':      # chunk 1 was prepared (reads 4..7) before chunk 0's forward was collected: the read of frame 4 precedes nothing that needs chunk 0's result
            assert log.index(('read', 4)) < log.index(('forward', 4))
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(outs['1'], outs['0'])) and len(outs['1']) == 10
    assert all(float(outs['1'][i][0][0, 0]) == float(i) for i in range(10))
    monkeypatch.setenv('KEEP_AMD_DETECT_OVERLAP', '1')
    proc = KEEPFaceProcessor(KEEPModelPack(_RecordingNet(), Helper(Det(fail_at=2)), None, None, 'KEEP'))
    with pytest.raises(RuntimeError, match='detector down'):
        proc._detect_all(frames, True)


def test_frames_from_comfy_host_fallbacks():
    """modules/keep_processor.py:frames_from_comfy without a GPU: a CPU device (and, on any device, inputs the device kernel does not take)
    yield the per-frame host converter's frames as a plain list -- the reference's `comfy_image_to_cv2` per frame."""
    from comfyui_keep_amd.modules import keep_processor as KPm
    from comfyui_keep_amd.modules.utils import comfy_image_to_cv2
    seq = torch.rand((3, 8, 10, 3), generator=torch.Generator().manual_seed(0))
    got = KPm.frames_from_comfy(seq, 'cpu')
    assert isinstance(got, list) and len(got) == 3
    assert all(np.array_equal(g, comfy_image_to_cv2(seq[i])) and g.dtype == np.uint8 and g.shape == (8, 10, 3) for i, g in enumerate(got))
    assert isinstance(KPm.frames_from_comfy(seq.double(), 'cuda'), list)          # not float32: the host converter, whatever the device


def test_device_crop_warp_hook_keeps_the_helper_contract(monkeypatch):
    """8f-2 host logic: with the GPU cv path on, ``_align_warp`` fits the similarity on the host (cv2.estimateAffinePartial2D) and
    crops on the device, filling ``affine_matrices`` / ``cropped_faces`` exactly like ``align_warp_face``; pad_blur or the cv path
    being off defers to the helper."""
    from comfyui_keep_amd.modules import keep_processor as KPm
    from comfyui_keep_amd.engine import paste as gp
    calls = []

    class Cv2:
        LMEDS = 4

        @staticmethod
        def estimateAffinePartial2D(lm, tmpl, method=None):
            return (np.array([[1.0, 0.0, float(lm[0, 0])], [0.0, 1.0, 0.0]]), None)

    class Hp(_Helper):
        pad_blur, face_size, face_template = False, (512, 512), np.zeros((5, 2))

        def __init__(self):
            self.affine_matrices, self.cropped_faces = [], []
            self.all_landmarks_5 = [np.full((5, 2), 3.0), np.full((5, 2), 7.0)]
            self.input_img = np.zeros((64, 64, 3), np.uint8)

        def align_warp_face(self):
            calls.append('helper')

    monkeypatch.setattr(KPm, '_cv2', lambda: Cv2)
    monkeypatch.setattr(gp, 'crop_faces', lambda frame, mats, size, dev: torch.zeros((len(mats), size[1], size[0], 3), dtype=torch.uint8))
    h = Hp()
    proc = KEEPFaceProcessor(KEEPModelPack(_RecordingNet(), h, None, None, 'KEEP'))
    proc.gpu_paste = True
    proc._align_warp(h)
    assert calls == [] and len(h.cropped_faces) == 2 and h.cropped_faces[0].shape == (512, 512, 3)
    assert [float(m[0, 2]) for m in h.affine_matrices] == [3.0, 7.0]
    h.pad_blur = True
    proc._align_warp(h)
    proc.gpu_paste, h.pad_blur = False, False
    proc._align_warp(h)
    assert calls == ['helper', 'helper']


def test_loader_swaps_the_helpers_networks_for_engine_objects(monkeypatch):
    """SURVEY 8f-4 host logic: ``engine_facelib`` replaces a ParseNet-shaped ``face_parse`` and a RetinaFace(resnet50 | mobile0.25)-shaped
    ``face_detector`` by engine-backed objects with the same call surface (weights packed on the host, nothing uploaded until
    ``.to('cuda')``), leaves other detectors alone, and KEEP_AMD_ENGINE_FACELIB=0 switches the swap off."""
    from comfyui_keep_amd.modules.keep_model_loader import engine_facelib
    from comfyui_keep_amd.engine import parsenet as PN, retinaface as RF

    class FakeModule:
        def __init__(self, sd, **attrs):
            self._sd = sd
            self.__dict__.update(attrs)

        def state_dict(self):
            return self._sd

    class Hp:
        pass
    h = Hp()
    h.face_parse = FakeModule(PN.synth_parsenet_state_dict(seed=0, in_size=128, out_size=128))
    h.face_detector = FakeModule(RF.synth_retinaface_state_dict(seed=0), backbone='Resnet50')
    engine_facelib(h)
    assert isinstance(h.face_parse, PN.EngineFaceParse) and (h.face_parse.engine.in_size, h.face_parse.engine.out_size) == (128, 128)
    assert isinstance(h.face_detector, RF.EngineRetinaFace) and h.face_detector.engine.w is None      # packed, not uploaded
    assert callable(h.face_detector.detect_faces) and callable(h.face_detector.detect_batch)
    with pytest.raises(RuntimeError):
        h.face_parse.engine.logits_nhwc(torch.zeros(1, 128, 128, 3))                                   # loud: not on a device
    h2 = Hp()
    h2.face_parse = object()                                    # not a ParseNet: untouched
    h2.face_detector = FakeModule({}, backbone='YOLOv5')       # no engine counterpart: untouched
    det2 = h2.face_detector
    engine_facelib(h2)
    assert h2.face_detector is det2 and not isinstance(h2.face_parse, PN.EngineFaceParse)
    hm = Hp()                                                   # retinaface_mobile0.25 (detection/__init__.py:38-41)
    hm.face_detector = FakeModule(RF.synth_retinaface_state_dict(seed=0, backbone='mobile0.25'), backbone='mobilenet0.25')
    engine_facelib(hm)
    assert isinstance(hm.face_detector, RF.EngineRetinaFace) and hm.face_detector.engine.backbone == 'mobile0.25'
    assert hm.face_detector.backbone == 'mobilenet0.25' and hm.face_detector.engine.w is None
    hy = Hp()                                                   # YOLOv5n / YOLOv5l: the YoloDetector keeps its host code, its network is swapped
    from comfyui_keep_amd.engine import yoloface as YF

    class FakeYolo:
        def __init__(self, sd):
            self.detector = FakeModule(sd)
    hy.face_detector = FakeYolo(YF.synth_yolo_state_dict('YOLOv5n', seed=0))
    engine_facelib(hy)
    assert isinstance(hy.face_detector, FakeYolo) and isinstance(hy.face_detector.detector, YF.EngineYoloModel)
    assert hy.face_detector.detector.engine.name == 'YOLOv5n' and float(hy.face_detector.detector.stride.max()) == 32.0
    assert not hasattr(hy.face_detector, 'detect_batch')      # (this fake has no _preprocess / _postprocess: per-frame path)
    # the batched pre-pass of a YoloDetector: its own _preprocess / _postprocess around ONE network call, detect_faces' rows per frame
    import types
    cv = types.ModuleType('cv2')
    cv.COLOR_BGR2RGB = 4
    cv.cvtColor = lambda img, code: img[..., ::-1]
    monkeypatch.setitem(sys.modules, 'cv2', cv)
    calls = []

    class FakeYolo2(FakeYolo):
        def _preprocess(self, images):
            calls.append(('pre', len(images), images[0][0, 0].tolist()))
            return torch.zeros(len(images), 3, 32, 32)

        def _postprocess(self, x, origimgs, pred, conf, iou):
            calls.append(('post', conf, iou, tuple(pred.shape)))
            return [[[1, 2, 30, 40]], [], [[5, 6, 7, 80], [9, 9, 20, 30]]], [[[[1, 1]] * 5], [], [[[2, 3]] * 5, [[4, 5]] * 5]]
    hy2 = Hp()
    hy2.face_detector = FakeYolo2(YF.synth_yolo_state_dict('YOLOv5n', seed=0))
    engine_facelib(hy2)
    hy2.face_detector.detector = lambda x: (torch.ones(x.shape[0], 7, 16), None)      # (no GPU here: the network itself is test_gpu_facelib's)
    frames = np.zeros((3, 8, 8, 3), np.uint8)
    frames[..., 0] = 9
    res = hy2.face_detector.detect_batch(frames, 0.97)
    assert calls == [('pre', 3, [0, 0, 9]), ('post', 0.97, 0.5, (3, 7, 16))]
    assert res[1] is None and res[0].shape == (1, 15) and res[2].shape == (2, 15)
    assert res[2][1].tolist() == [9, 9, 20, 30, 9, 4, 5, 4, 5, 4, 5, 4, 5, 4, 5]
    assert set(YF.synth_yolo_state_dict('YOLOv5l', seed=0)) == set(YF.yolo_state_dict_spec('YOLOv5l')) and YF.config_of(YF.yolo_state_dict_spec('YOLOv5l')) == 'YOLOv5l'
    with pytest.raises(RuntimeError):                           # a resnet50 trunk under the mobile name: loud
        RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0), backbone='mobile0.25')
    monkeypatch.setenv('KEEP_AMD_ENGINE_FACELIB', '0')
    h3 = Hp()
    h3.face_parse = FakeModule(PN.synth_parsenet_state_dict(seed=0, in_size=128, out_size=128))
    engine_facelib(h3)
    assert isinstance(h3.face_parse, FakeModule)
