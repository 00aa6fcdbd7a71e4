"""CPU suite: the oracle (oracle/keep_oracle.py) against the golden vectors generated from the imported
reference (oracle/make_golden.py).  This is what pins the oracle; tolerance covers fp32 re-association
between the reference's module graph and the functional restatement (measured 6e-5 on full forwards)."""
import json
import os

import numpy as np
import pytest
import torch

import keep_oracle as O
from conftest import GOLDEN, op_input
from comfyui_keep_amd.engine import arch, synth

OPS = np.load(os.path.join(GOLDEN, 'ops.npz'))
TOL = 2e-4


def close(a, b, tol=TOL):
    a = a.numpy() if torch.is_tensor(a) else a
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), f"max abs err {err}"


def test_arch_spec_matches_reference_listing():
    with open(os.path.join(GOLDEN, 'arch_spec.json')) as f:
        ref = json.load(f)
    for name, over in (('KEEP', {}), ('Asian', {'cft_list': ['32', '64', '128', '256'], 'temp_reg_list': []})):
        spec = arch.state_dict_spec(dict(arch.DEFAULT_ARCH, **over))
        assert set(spec) == set(ref[name])
        assert all(list(spec[k]) == ref[name][k] for k in spec)
    assert len(ref['KEEP']) == 896


def test_resblocks(synth_weights):
    W = synth_weights
    close(O.resblock(op_input('res_same', (1, 128, 16, 16)), W, 'encoder.blocks.5'), OPS['res_same'])
    close(O.resblock(op_input('res_proj', (1, 64, 16, 16)), W, 'encoder.blocks.4'), OPS['res_proj'])


def test_down_up_attn(synth_weights):
    W = synth_weights
    close(O.downsample(op_input('down', (1, 128, 16, 16)), W, 'encoder.blocks.6'), OPS['down'])
    close(O.upsample(op_input('up', (1, 128, 8, 8)), W, 'generator.blocks.17'), OPS['up'])
    close(O.attnblock(op_input('attn', (1, 512, 8, 8)), W, 'encoder.blocks.17'), OPS['attn'])


def test_transformer_layer(synth_weights):
    W = synth_weights
    pos = W['position_emb'][:64].unsqueeze(1)
    close(O.transformer_sa_layer(op_input('sa_layer', (64, 1, 512)), pos, W, 'ft_layers.0', 8), OPS['sa_layer'])


def test_codebook_and_nearest(synth_weights):
    W = synth_weights
    idx = torch.from_numpy(((synth.uniform_pm1('op_input:codes', 64, 7) + 1) * 512).astype(np.int64).clip(0, 1023))
    assert np.array_equal(O.codebook_lookup(idx.view(1, 64), W, 1, 8, 256).numpy(), OPS['codebook'])
    nn = O.nearest_codes(op_input('vq_nn', (1, 256, 8, 8), 0.7), W)
    assert np.array_equal(nn.numpy().astype(np.int32), OPS['vq_nn_idx'])


def test_cft_cfa(synth_weights):
    W = synth_weights
    close(O.cft_fuse(op_input('cft_enc', (1, 256, 8, 8)), op_input('cft_dec', (1, 256, 8, 8)), W, 'cft.32', 1), OPS['cft'])
    close(O.cfa_fuse(op_input('cfa_curr', (1, 256, 8, 8)), op_input('cfa_prev', (1, 256, 8, 8)), W, 'cfa.32', 4, 256),
          OPS['cfa'])


def test_kalman(synth_weights):
    W = synth_weights
    close(O.kalman_calc_gain(op_input('kalman_z', (1, 3, 256, 8, 8)), W, arch.DEFAULT_ARCH), OPS['kalman_gain'])
    g = (op_input('ku_g', (1, 1, 8, 8)) + 1) / 2
    zc, zp = op_input('ku_z', (1, 256, 8, 8)), op_input('ku_zp', (1, 256, 8, 8))
    close((1 - g) * zc + g * zp, OPS['kalman_update'], 1e-6)


def test_flow_warp():
    close(O.flow_warp(op_input('warp_img', (2, 3, 32, 32)), op_input('warp_flow', (2, 32, 32, 2), 6.0)), OPS['warp'], 1e-6)


def test_gmflow64(synth_weights):
    a = synth.synth_clip(T=2, B=1, size=64, seed=99)[0]
    close(O.gmflow_forward(a[1:2], a[0:1], synth_weights), OPS['gmflow64'])


def _flow_report(flow, ref):
    """Error of a flow field against the reference's, absolute (px) and against the flow scale: max, and the 99th percentile of
    the per-pixel error relative to max(1 px, that pixel's reference flow)."""
    err = np.abs(flow - ref)
    mag = np.maximum(1.0, np.sqrt((ref ** 2).sum(1, keepdims=True)))
    return {'max_err_px': float(err.max()), 'p99_rel': float(np.quantile(err / mag, 0.99)), 'median_px': float(np.median(np.sqrt((ref ** 2).sum(1)))),
            'scale_px': float(np.abs(ref).max())}


def test_gmflow256_physical_regime(synth_weights):
    """M16-M21 pinned where a relative bound means something: the reference's FlowGenerator on frames 0 and 3 of
    the translating texture at 256x256 (|flow| median 0.87 px, p99 6.9 px, max 88 px), EVERY pixel of the field."""
    g = np.load(os.path.join(GOLDEN, 'gmflow256.npz'))
    dt = int(g['dt'])
    a = synth.synth_clip(T=dt + 1, B=1, size=256, seed=int(g['clip_seed']))[0]
    flow = O.gmflow_forward(a[dt:dt + 1], a[0:1], synth_weights).numpy()
    rep = _flow_report(flow, g['flow'])
    print('oracle gmflow256 vs reference', rep)
    assert 0.5 < rep['median_px'] < 2.0
    assert rep['max_err_px'] <= 2e-4 * rep['scale_px'] and rep['p99_rel'] <= 2e-4, rep


def test_encoder_generator_small(synth_weights):
    z, _ = O.encoder_forward(op_input('encoder64', (1, 3, 64, 64)), synth_weights, 'encoder', arch.DEFAULT_ARCH)
    close(z, OPS['encoder64'])
    y = op_input('generator_2x2', (1, 256, 2, 2), 0.7)
    for j, (kind, _, _) in enumerate(arch.generator_blocks(arch.DEFAULT_ARCH)):
        y = O.vq_block(y, synth_weights, f'generator.blocks.{j}', kind)
    close(y, OPS['generator_2x2'])


def _digest(frames):
    T, C, H, Wd = frames.shape
    return frames[:, :, 7::H // 32, 5::Wd // 32][:, :, :32, :32].numpy()


def test_full_forward_T2_vs_golden(synth_weights):
    """First two frames of the golden T=3 clip (frame i depends only on frames <= i ... except the Kalman gain,
    which attends over the whole clip; so compare indices/outputs of a genuine T=3 run only in the slow test and
    here check the T-independent frame 0)."""
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T3.npz'))
    x = synth.synth_clip(T=2, B=1, seed=1234)
    out, aux = O.keep_forward(x, synth_weights, return_aux=True)
    assert np.array_equal(aux['indices'][0, 0].numpy().astype(np.int16), g['indices'][0])
    err = np.abs(_digest(out[0])[0] - g['out_grid'][0]).max()
    assert err <= 2e-4, err


@pytest.mark.slow
def test_full_forward_T3_vs_golden(synth_weights):
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T3.npz'))
    x = synth.synth_clip(T=3, B=1, seed=1234)
    out, aux = O.keep_forward(x, synth_weights, return_aux=True)
    safe = g['margins'] > 1e-3
    assert np.array_equal(aux['indices'][0].numpy().astype(np.int16)[safe], g['indices'][safe])
    assert np.abs(aux['gains'][0, :, 0].reshape(3, -1).numpy() - g['gains']).max() <= 1e-4
    assert np.abs(_digest(out[0]) - g['out_grid']).max() <= 3e-4


@pytest.mark.slow
def test_full_forward_T3_every_pixel_of_the_centre_crop_vs_reference(synth_weights):
    """tests/golden/keep_forward_T3_pixels.npz: every pixel of the 128x128 centre crop of all three frames, from the imported
    reference itself (oracle/make_golden_r5.py)."""
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T3_pixels.npz'))
    a, b, c, d = (int(v) for v in g['crop'])
    x = synth.synth_clip(T=3, B=1, seed=1234)
    out = O.keep_forward(x, synth_weights)
    err = np.abs(out[0][:, :, a:b, c:d].numpy() - g['out_crop']).reshape(3, -1).max(1)
    print('oracle vs reference, every pixel of the centre crop, per frame:', err)
    assert err.max() <= 3e-4, err


@pytest.mark.slow
def test_full_forward_T3_wide_flow_regime_vs_golden():
    """Round 3's regime (i.i.d. flownet weights + the plane-wave clip: flows of hundreds of pixels, most warp samples outside the
    frame) kept as the out-of-range edge case: tests/golden/keep_forward_T3_wide.npz."""
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T3_wide.npz'))
    assert np.median(np.sqrt((g['flow_grid'] ** 2).sum(1))) > 50.0
    W = synth.synth_state_dict(DEFAULT_ARCH, seed=0, flow_regime='wide')
    x = synth.synth_clip(T=3, B=1, seed=1234, pattern='waves')
    out, aux = O.keep_forward(x, W, return_aux=True)
    safe = g['margins'] > 1e-2
    assert np.array_equal(aux['indices'][0].numpy().astype(np.int16)[safe], g['indices'][safe])
    assert np.abs(aux['gains'][0, :, 0].reshape(3, -1).numpy() - g['gains']).max() <= 1e-4
    assert np.abs(_digest(out[0])[0] - g['out_grid'][0]).max() <= 3e-4


@pytest.mark.slow
def test_full_forward_T20_vs_golden(synth_weights):
    """The metric's own clip length on the CPU oracle against the imported reference (tests/golden/keep_forward_T20.npz):
    code indices wherever the reference's margin exceeds 1e-3, frame by frame up to the first frame with any differing
    token (beyond it the two runs restore different inputs), gains, and every digest pixel of those frames within an
    ABSOLUTE 1e-3 (the synthetic net's frames live in [-1.3, 1])."""
    g = np.load(os.path.join(GOLDEN, 'keep_forward_T20.npz'))
    x = synth.synth_clip(T=20, B=1, seed=1234)
    out, aux = O.keep_forward(x, synth_weights, return_aux=True)
    idx = aux['indices'][0].numpy().astype(np.int16)
    agree = idx == g['indices']
    first = next((t for t in range(20) if not agree[t].all()), 20)
    assert np.abs(aux['gains'][0, :, 0].reshape(20, -1).numpy() - g['gains']).max() <= 1e-4
    for t in range(min(first + 1, 20)):
        assert agree[t][g['margins'][t] > 1e-3].all(), t
    assert first >= 1
    err = np.abs(_digest(out[0])[:first] - g['out_grid'][:first]).max()
    print('oracle vs reference, T=20: first frame with a differing index', first, '; max-abs digest diff before it', err)
    assert err <= 1e-3, err
    # round 6: every pixel of the 128x128 centre crop of frames 0-3 of the same reference run (keep_forward_T20_pixels.npz,
    # oracle/make_golden_r6.py)
    gp = np.load(os.path.join(GOLDEN, 'keep_forward_T20_pixels.npz'))
    a, b, c, d = (int(v) for v in gp['crop'])
    frames = [int(v) for v in gp['frames']]
    assert first > max(frames)
    perr = np.abs(out[0][frames][:, :, a:b, c:d].numpy() - gp['out_crop']).reshape(len(frames), -1).max(1)
    print('oracle vs reference, T=20, every pixel of the centre crop of frames', frames, ':', perr)
    assert perr.max() <= 3e-4, perr


def test_converters_oracle_and_host_converters_vs_reference_golden():
    """P4: oracle/converters_oracle.py against the reference's own img2tensor / tensor2img (img_util.py:9-94, called as
    keep_processor.py:258-259,272 do; golden ``conv_in_crops`` / ``conv_out_u8``), bit for bit; the product's host converters
    (modules/utils.py) against the oracle."""
    import converters_oracle as CO
    from comfyui_keep_amd.modules import utils as U
    g = np.load(os.path.join(GOLDEN, 'ops.npz'))
    crops = [synth.ramp_image(64, 64), np.ascontiguousarray(synth.ramp_image(64, 64)[::-1])]
    x_in = CO.crops_to_net_input(crops)
    assert x_in.dtype == np.float32 and np.array_equal(x_in, g['conv_in_crops'])
    assert torch.equal(U.crops_to_net_input(crops), torch.from_numpy(x_in))
    xo = op_input('t2i', (2, 3, 64, 64), 1.3)
    xo[0, :, 0, :8] = torch.tensor([-1.2, -1.0, -0.5 / 255, 0.0, 1.0 / 255, 1.0, 1.3, 0.00392156862])
    for n in range(2):
        got = CO.net_output_to_bgr_u8(xo[n].numpy())
        assert got.dtype == np.uint8 and np.array_equal(got, g['conv_out_u8'][n])
        assert np.array_equal(U.net_output_to_bgr_u8(xo[n]), got)


def test_parsenet_oracle_vs_reference_golden():
    """oracle/facelib_oracle.py:parsenet_forward against the imported reference ParseNet (tests/golden/facelib.npz)."""
    import facelib_oracle as FO
    from comfyui_keep_amd.engine import parsenet as PN
    g = np.load(os.path.join(GOLDEN, 'facelib.npz'))
    W = PN.synth_parsenet_state_dict(seed=0, in_size=128, out_size=128)
    x = op_input('parsenet128', (2, 3, 128, 128))
    with torch.no_grad():
        mask = FO.parsenet_forward(x, W, PN.parsenet_spec(in_size=128, out_size=128))
    assert mask.shape == (2, 19, 128, 128)
    assert np.abs(mask[:, :, 1::4, 2::4].numpy() - g['parsenet128_logit_grid']).max() <= 2e-4 * np.abs(g['parsenet128_logit_grid']).max()
    safe = g['parsenet128_margin'].astype(np.float32) > 1e-2
    assert np.array_equal(mask.argmax(1).numpy().astype(np.uint8)[safe], g['parsenet128_classes'][safe]) and safe.mean() > 0.99


def test_retinaface_oracle_and_host_decode_vs_reference_golden():
    """oracle/facelib_oracle.py:retinaface_forward against the reference's own FPN / SSH / head modules composed over the same
    (restated, unpinned) ResNet-50 trunk; engine/retinaface.py's prior boxes and decoders against the reference's PriorBox /
    decode / decode_landm (tests/golden/facelib.npz)."""
    import facelib_oracle as FO
    from comfyui_keep_amd.engine import retinaface as RF
    g = np.load(os.path.join(GOLDEN, 'facelib.npz'))
    W = RF.synth_retinaface_state_dict(seed=0)
    assert set(W) == set(RF.retinaface_state_dict_spec())
    x = op_input('retinaface_img', (2, 3, 160, 224), 100.0)
    with torch.no_grad():
        loc, conf, lm = FO.retinaface_forward(x, W)
    for got, key in ((loc, 'retinaface_loc'), (conf, 'retinaface_conf'), (lm, 'retinaface_landm')):
        assert np.abs(got.numpy() - g[key]).max() <= 1e-4 * max(1.0, np.abs(g[key]).max()), key
    pri = RF.prior_boxes(160, 224)
    assert pri.shape == g['retinaface_priors'].shape and np.abs(pri - g['retinaface_priors']).max() <= 1e-7
    assert np.abs(RF.decode_boxes(g['retinaface_loc'][0], pri, RF.CFG_RE50['variance']) - g['retinaface_boxes0']).max() <= 1e-5
    assert np.abs(RF.decode_landmarks(g['retinaface_landm'][0], pri, RF.CFG_RE50['variance']) - g['retinaface_lms0']).max() <= 1e-5
    # mobile0.25: the reference's OWN MobileNetV1 + FPN + SSH + heads from image to heads (no unpinned part)
    Wm = RF.synth_retinaface_state_dict(seed=0, backbone='mobile0.25')
    assert set(Wm) == set(RF.retinaface_state_dict_spec('mobile0.25')) and RF.backbone_of(Wm) == 'mobile0.25' and RF.backbone_of(W) == 'resnet50'
    xm = op_input('retinaface_mnet_img', (2, 3, 160, 224), 100.0)
    with torch.no_grad():
        feats = FO.mobilenet_trunk(xm, Wm)
        loc, conf, lm = FO.retinaface_forward(xm, Wm, backbone='mobile0.25')
    grid = torch.cat([f[:, ::8, ::3, ::5].reshape(-1) for f in feats]).numpy()
    assert [f.shape[1] for f in feats] == [64, 128, 256] and np.abs(grid - g['mnet_stage_grid']).max() <= 1e-5
    for got, key in ((loc, 'mnet_loc'), (conf, 'mnet_conf'), (lm, 'mnet_landm')):
        assert np.abs(got.numpy() - g[key]).max() <= 1e-4 * max(1.0, np.abs(g[key]).max()), key
    assert np.abs(RF.decode_boxes(g['mnet_loc'][0], pri, RF.CFG_MNET['variance']) - g['mnet_boxes0']).max() <= 1e-5
    assert np.abs(RF.decode_landmarks(g['mnet_landm'][0], pri, RF.CFG_MNET['variance']) - g['mnet_lms0']).max() <= 1e-5
    # YOLOv5n / YOLOv5l: the reference's own Model(yaml) from image to the decoded [N, anchors, 16] predictions
    from comfyui_keep_amd.engine import yoloface as YF
    for name in ('YOLOv5n', 'YOLOv5l'):
        Wy = YF.synth_yolo_state_dict(name, seed=0)
        xy = op_input(f'yolo_img_{name}', (2, 3, 96, 128)).mul(0.5).add(0.5).clamp(0, 1)
        with torch.no_grad():
            pred = FO.yolo_forward(xy, Wy, YF.yolo_layers(name), YF.ANCHORS, YF.STRIDES)
        ref = g[f'{name.lower()}_pred']
        assert pred.shape == ref.shape and np.abs(pred.numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), name
    # greedy NMS: a cluster of overlapping boxes keeps its best, disjoint boxes all survive, order = descending score
    d = np.array([[0, 0, 10, 10, 0.9], [1, 1, 11, 11, 0.95], [20, 20, 30, 30, 0.5], [0, 0, 10, 10.5, 0.7]], np.float32)
    assert RF.nms(d, 0.4) == [1, 2]


def test_yolo_pre_and_post_processing_vs_reference_golden():
    """oracle/facelib_oracle.py:yolo_preprocess / yolo_postprocess and the product's host logic (engine/yoloface.py:letterbox_geometry,
    faces_from_kept_rows) against what the reference's OWN ``YoloDetector.detect_faces`` returned frame by frame
    (tests/golden/yolo_prepost.npz; cv2.resize / torchvision.ops.nms unpinned, see the oracle's header): the network input it built and
    the face rows it assembled, at detect_faces' default thresholds and at the helper's 0.97."""
    import facelib_oracle as FO
    from comfyui_keep_amd.engine import yoloface as YF
    g = np.load(os.path.join(GOLDEN, 'yolo_prepost.npz'))
    for hw in ((100, 160), (90, 150), (640, 1137), (1080, 1920), (512, 512), (720, 1280), (333, 517), (517, 333), (64, 2000)):
        assert YF.letterbox_geometry(*hw) == FO.yolo_letterbox_geometry(*hw), hw
    for tag, (frames, pred) in FO.yolo_prepost_inputs().items():
        (rh, rw), (top, left), net_hw = YF.letterbox_geometry(*frames.shape[1:3])
        x = FO.yolo_preprocess(list(frames))
        assert tuple(x.shape[2:]) == net_hw == g[f'{tag}_x0'].shape[1:]
        for n in range(2):
            assert np.array_equal(x[n].numpy(), g[f'{tag}_x{n}'].astype(np.float32) / np.float32(255.0)), (tag, n)
            if tag == 'pad':            # no resize on this frame size: the golden input has no unpinned part -- it IS the frame, RGB, inside 114
                assert np.array_equal(g[f'{tag}_x{n}'][:, top:top + rh, left:left + rw], frames[n][:, :, ::-1].transpose(2, 0, 1))
                assert (g[f'{tag}_x{n}'][:, :top] == 114).all() and (g[f'{tag}_x{n}'][:, top + rh:] == 114).all()
            for conf, name in ((0.7, 'default'), (0.97, 'helper')):
                want = g[f'{tag}_{name}_det{n}']
                got = FO.yolo_postprocess(pred[n], net_hw, frames.shape[1:3], conf, 0.5)
                assert np.array_equal(np.zeros((0, 15), np.int64) if got is None else got, want), (tag, name, n)
                # the product's host tail on the rows the selection + suppression leave (restated here with the oracle's NMS)
                p = pred[n][pred[n][:, 4] > np.float32(conf)]
                c = p[:, 15] * p[:, 4]
                rows = np.concatenate((np.stack((p[:, 0] - p[:, 2] / 2, p[:, 1] - p[:, 3] / 2, p[:, 0] + p[:, 2] / 2, p[:, 1] + p[:, 3] / 2, c), 1),
                                       p[:, 5:15]), 1)[c > np.float32(conf)]
                keep = FO._greedy_nms(rows[:, :4], rows[:, 4], np.float32(0.5)) if len(rows) else []
                got = YF.faces_from_kept_rows(rows[keep] if keep else None, net_hw, frames.shape[1:3], 10)
                assert np.array_equal(np.zeros((0, 15), np.int64) if got is None else got, want), (tag, name, n, 'host tail')
    # edge cases of the product's host tail: no rows, only faces under min_face, clamps at the frame border, truncation toward zero
    assert YF.faces_from_kept_rows(None, (160, 160), (100, 160), 10) is None
    assert YF.faces_from_kept_rows(np.zeros((0, 16), np.float32), (160, 160), (100, 160), 10) is None
    tiny = np.array([[50.0, 60.0, 58.0, 69.9, 0.99] + [55.0, 65.0] * 5 + [0.0]], np.float32)          # 9.9 px high after the 30 px of padding go
    assert YF.faces_from_kept_rows(tiny, (160, 160), (100, 160), 10) is None
    edge = np.array([[-5.0, 20.0, 170.0, 140.0, 0.99] + [-3.0, 200.0] * 5 + [0.0]], np.float32)
    got = YF.faces_from_kept_rows(edge, (160, 160), (100, 160), 10)
    assert got.dtype == np.int64 and got.tolist() == [[0, 0, 160, 100, 0] + [0, 100] * 5]
    ref = FO.yolo_postprocess(np.concatenate((np.array([[82.5, 80.0, 175.0, 120.0, 0.99]], np.float32), edge[:, 5:15], np.array([[1.0]], np.float32)), 1),
                              (160, 160), (100, 160), 0.7, 0.5)
    assert np.array_equal(ref, got)
    # the resize restatement: identity, constant images, and a 2 x 2 -> 4 x 4 case by hand (coefficients 2048 * {.75, .25})
    img = np.arange(2 * 2 * 3, dtype=np.uint8).reshape(2, 2, 3) * 20
    up = FO.cv2_resize_linear_u8(img, 4, 4)
    assert np.array_equal(up[0, 0], img[0, 0]) and np.array_equal(up[3, 3], img[1, 1])
    hor = (int(img[0, 0, 0]) * 1536 + int(img[0, 1, 0]) * 512) >> 4          # dx = 1: fx = .25; dy = 0: sy = -1 (clamped to row 0), fy = .75
    assert int(up[0, 1, 0]) == (((512 * hor) >> 16) + ((1536 * hor) >> 16) + 2) >> 2
    assert (FO.cv2_resize_linear_u8(np.full((7, 9, 3), 93, np.uint8), 20, 13) == 93).all()
    with pytest.raises(NotImplementedError):
        FO.cv2_resize_linear_u8(np.zeros((8, 8, 3), np.uint8), 4, 4)
