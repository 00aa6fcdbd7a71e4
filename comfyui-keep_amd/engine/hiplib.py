"""ctypes binding of ``libkeep_hip.so`` (the C-ABI declared in ``include/keep_hip.h``).

This is the host side of the drop-in boundary: plain pointers and sizes cross it, torch only
supplies device memory (``tensor.data_ptr()``) and the stream.  There is NO fallback: if the
library is missing, was built for another ABI version, or the device is not gfx950, loading
raises -- the product path never silently runs on anything but the hand-written kernels.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KEEP_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libkeep_hip.so')   # (KEEP_HIP_LIB: dev A/B builds)
ABI_VERSION = 20

F32, BF16 = 0, 1
MMA_F32, MMA_BF16, MMA_X3 = 0, 1, 2
PRO_NONE, PRO_SWISH, PRO_RELU = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_GELU, ACT_SIGMOID, ACT_LRELU01, ACT_SILU = 0, 1, 2, 3, 4, 5, 6
PAD_ZERO, PAD_REFLECT = 0, 1
UPSAMPLE_X2_PHASES = 2
# keep_conv2d_args.flags / keep_attention_args.flags (include/keep_hip.h): kernel-selection overrides, 0 = the library's choice
(CONV_NO_COUT4, CONV_NO_C3, CONV_NO_HALO_F32, CONV_NO_HALO_X3, CONV_NO_GATHER_X3, CONV_NO_PLAIN, CONV_NO_FLATK_F32,
 CONV_SMALL_TILES, CONV_X3_EXACT_ACT, CONV_NO_STREAM, CONV_NO_GEMM_LAT, CONV_GEMM_LAT_WAVES, CONV_GEMM_LAT_TILES, CONV_NO_SMALL_PARTIALS) = (1 << i for i in range(14))
ATTN_NO_PACK, ATTN_NO_SFULL2, ATTN_NO_X3, ATTN_NO_SMALL, ATTN_NO_TWO_PASS = 1, 2, 4, 8, 16

_vp, _i32, _i64, _f32, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32
STATUS_NONFINITE_LOGITS, STATUS_NONFINITE_TENSOR = 1, 2


class ConvArgs(C.Structure):
    # field-for-field include/keep_hip.h:keep_conv2d_args ('inp' = `in`, a Python keyword); tests/test_host_logic.py checks the
    # names / order against the header, the size against keep_sizeof_conv2d_args(), and INTEGRATION.md's copy against this list
    _fields_ = [('struct_size', _u32), ('flags', _u32), ('inp', _vp), ('weight', _vp), ('bias', _vp), ('out', _vp), ('pro_scale', _vp), ('pro_shift', _vp),
                ('residual', _vp), ('aux', _vp), ('workspace', _vp)] + \
               [(n, _i32) for n in ('N', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'pad_t', 'pad_l', 'Ho', 'Wo',
                                    'in_ld', 'out_ld', 'res_ld', 'upsample', 'pro_act', 'epi_act')] + \
               [('aux_w', _f32), ('split_k', _i32), ('dtype', _i32), ('mma', _i32), ('weight_bf16', _vp), ('stats_out', _vp), ('stats_P', _i32), ('bk256', _i32), ('out_dtype', _i32),
                ('weight_x3', _vp), ('x3_acc_scale', _f32), ('x3_in_amax', _vp), ('x3_out_amax', _vp), ('x3_out_amax_zeroed', _i32), ('in2', _vp), ('in2_cin1', _i32), ('pad_mode', _i32),
                ('ln_gamma', _vp), ('ln_beta', _vp), ('ln_eps', _f32), ('plan_ref_images', _i32)]


class ConvPlanOut(C.Structure):
    _fields_ = [('path', _i32), ('split_k', _i32), ('workspace_bytes', _i64), ('stats_rows', _i32), ('stats_P', _i32),
                ('wants_bf16_input', _i32), ('out_bf16_ok', _i32), ('out_amax_ok', _i32), ('kernel', C.c_char * 64)]


class AttnArgs(C.Structure):
    _fields_ = [('struct_size', _u32), ('flags', _u32), ('q', _vp), ('k', _vp), ('v', _vp), ('o', _vp)] + \
               [(n, _i64) for n in ('q_bs', 'q_ts', 'q_hs', 'k_bs', 'k_ts', 'k_hs', 'v_bs', 'v_ts', 'v_hs',
                                    'o_bs', 'o_ts', 'o_hs')] + \
               [(n, _i32) for n in ('B', 'H', 'Lq', 'Lk', 'D', 'Dv')] + [('scale', _f32), ('mode', _i32)] + \
               [(n, _i32) for n in ('T', 'seg_len', 'img_h', 'img_w', 'ksplit', 'shift', 'kv_rot', 'n_img', 'mma', 'in_dtype')] + \
               [('q_amax', _vp), ('k_amax', _vp), ('v_amax', _vp), ('workspace', _vp), ('workspace_bytes', _i64)]


# name -> argtypes (restype is always int32 status); every symbol include/keep_hip.h declares
_SIGNATURES = {
    'keep_conv2d': [C.POINTER(ConvArgs), _vp],
    'keep_conv2d_plan': [C.POINTER(ConvArgs), C.POINTER(ConvPlanOut)],
    'keep_attention': [C.POINTER(AttnArgs), _vp],
    'keep_chan_stats': [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_norm_finalize': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    'keep_group_stats': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    'keep_affine_act': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_gm_mlp': [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    'keep_token_linear': [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    'keep_gm_ffn_x3': [_vp, _vp, _vp, _f32, _vp, _f32, _vp, _vp, _f32, _vp, _i64, _i32, _i32, _i32, _vp],
    'keep_norm_act_bf16': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_gm_join': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'keep_bilinear_upscale': [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_absmax': [_vp, _vp, _i32, _i64, _i32, _i64, _i64, _i32, _vp],
    'keep_layernorm': [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp],
    'keep_geglu': [_vp, _vp, _i32, _i32, _vp],
    'keep_layernorm_amax': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _i32, _vp],
    'keep_geglu_amax': [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    'keep_argmax_gather': [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    'keep_nonfinite_flag': [_vp, _i64, _vp, _vp],
    'keep_vq_nearest': [_vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'keep_kalman_update': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'keep_flow_warp': [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_convex_upsample': [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_nchw_to_nhwc': [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_rgb_s2d': [_vp, _vp, _i32, _i32, _i32, _vp],
    'keep_nhwc_to_nchw': [_vp, _vp, _i32, _i32, _i32, _vp],
    'keep_add_bcast': [_vp, _vp, _vp, _i64, _i64, _f32, _vp],
    'keep_concat2': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    'keep_tensor2img': [_vp, _vp, _i64, _vp],
    'keep_img2tensor': [_vp, _vp, _i64, _vp],
    'keep_bgr_u8_to_comfy': [_vp, _vp, _i64, _vp],
    'keep_comfy_to_bgr_u8': [_vp, _vp, _i64, _vp],
    'keep_channel_argmax': [_vp, _vp, _i64, _i32, _i32, _vp],
    'keep_maxpool3s2': [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_dwconv3x3': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_maxpool2d': [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_slice_copy': [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_channel_shuffle2': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    'keep_yolo_decode': [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _i32, _i32, _vp],
    'keep_yolo_letterbox_u8': [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_yolo_select': [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp],
    'keep_upsample_add': [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    'keep_act_inplace': [_vp, _i64, _i32, _vp],
    'keep_retina_decode': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _f32, _vp],
    'keep_retina_nms': [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp],
    'keep_retina_nms_ordered': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp],
    'keep_sep_filter': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    'keep_u8_to_f32': [_vp, _vp, _i64, _vp],
    'keep_f32_round_u8': [_vp, _vp, _i64, _vp],
    'keep_warp_affine_u8': [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    'keep_warp_ones': [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    'keep_draw_box': [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    'keep_erode_rect': [_vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'keep_paste_face': [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
}
EXPORTED_SYMBOLS = ['keep_abi_version', 'keep_last_error', 'keep_device_ok', 'keep_attention_workspace_bytes',
                    'keep_sizeof_conv2d_args', 'keep_sizeof_attention_args'] + list(_SIGNATURES)

_lib = None
_device_checked = set()


class KeepHipError(RuntimeError):
    pass


def load(check_device=True):
    """dlopen the library, bind every symbol, verify the ABI version (and gfx950 when asked)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KeepHipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                f"There is no CPU/PyTorch fallback for the KEEP hot path.")
        lib = C.CDLL(LIB_PATH)
        lib.keep_abi_version.restype = _i32
        lib.keep_last_error.restype = C.c_char_p
        lib.keep_device_ok.restype = _i32
        lib.keep_device_ok.argtypes = [_i32]
        lib.keep_attention_workspace_bytes.restype = _i64
        lib.keep_attention_workspace_bytes.argtypes = [C.POINTER(AttnArgs)]
        ver = lib.keep_abi_version()
        if ver != ABI_VERSION:
            raise KeepHipError(f"libkeep_hip.so ABI version {ver} != expected {ABI_VERSION}; rebuild")
        lib.keep_sizeof_conv2d_args.restype = lib.keep_sizeof_attention_args.restype = _i32
        for name, struct in (('keep_sizeof_conv2d_args', ConvArgs), ('keep_sizeof_attention_args', AttnArgs)):
            want = getattr(lib, name)()
            if want != C.sizeof(struct):        # a layout drift between this binding and the library is a load-time error
                raise KeepHipError(f"{name}() = {want} but the ctypes {struct.__name__} is {C.sizeof(struct)} bytes; "
                                   f"engine/hiplib.py and include/keep_hip.h disagree")
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(lib, name)            # AttributeError if the symbol is missing
            fn.restype = _i32
            fn.argtypes = argtypes
        _lib = lib
    if check_device:
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        if dev is None:
            raise KeepHipError("no HIP device visible: the KEEP hot path runs on MI355X (gfx950) only")
        if dev not in _device_checked:
            rc = _lib.keep_device_ok(dev)
            if rc != 0:
                raise KeepHipError(_lib.keep_last_error().decode())
            _device_checked.add(dev)
    return _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(rc, what):
    if rc != 0:
        raise KeepHipError(f"{what} failed (code {rc}): {_lib.keep_last_error().decode()}")


def call(name, *args):
    """Invoke a flat-signature entry point on torch's current stream; tensors are passed as pointers."""
    lib = load()
    conv = [(_ptr(a) if isinstance(a, torch.Tensor) or a is None else a) for a in args]
    _check(getattr(lib, name)(*conv, _stream()), name)


def conv_args(**kw):
    a = ConvArgs()
    a.struct_size = C.sizeof(ConvArgs)
    for k, v in kw.items():
        setattr(a, k, _ptr(v) if (isinstance(v, torch.Tensor) or v is None) else v)
    return a


def conv2d_plan(a):
    """keep_conv2d_plan: the library's own decision for these arguments (kernel, split-K, workspace / statistics sizes)."""
    lib = load(check_device=False)
    out = ConvPlanOut()
    _check(lib.keep_conv2d_plan(C.byref(a), C.byref(out)), 'keep_conv2d_plan')
    return out


def attention_workspace_bytes(a):
    """keep_attention_workspace_bytes: scratch the library can use for this call (0 = none)."""
    return int(load(check_device=False).keep_attention_workspace_bytes(C.byref(a)))


def conv2d_launch(a):
    _check(load().keep_conv2d(C.byref(a), _stream()), 'keep_conv2d')


def conv2d(**kw):
    conv2d_launch(conv_args(**kw))


def attention(**kw):
    lib = load()
    a = AttnArgs()
    a.struct_size = C.sizeof(AttnArgs)
    for k, v in kw.items():
        setattr(a, k, _ptr(v) if (isinstance(v, torch.Tensor) or v is None) else v)
    need = int(lib.keep_attention_workspace_bytes(C.byref(a)))       # the library asks; the host only allocates
    if need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=kw['q'].device)   # stream-ordered free after the call
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _check(lib.keep_attention(C.byref(a), _stream()), 'keep_attention')
