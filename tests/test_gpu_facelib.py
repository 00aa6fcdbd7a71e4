"""GPU suite (-m gpu): the face-analysis networks of SURVEY.md 8f-4 on the HIP engine against the golden vectors generated
from the imported reference modules (tests/golden/facelib.npz) and the CPU oracle (oracle/facelib_oracle.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import facelib_oracle as FO
from conftest import GOLDEN, op_input
from comfyui_keep_amd.engine import hiplib as L
from comfyui_keep_amd.engine import ops
from comfyui_keep_amd.engine import parsenet as PN

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, 'facelib.npz'))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


@pytest.mark.parametrize("mma", [L.MMA_F32, L.MMA_X3])
def test_reflect_padding_every_kernel_family(mma):
    """keep_conv2d pad_mode = KEEP_PAD_REFLECT (nn.ReflectionPad2d(1), parsenet.py:97) on every kernel family that takes it:
    LDS-halo (stride 1, with and without the nearest-x2 gather), gather (stride 2; ragged map), flattened-K (Cin = 3)."""
    cases = [('halo', 2, 64, 64, 32, 32, 1, False), ('halo up', 2, 64, 64, 16, 16, 1, True), ('gather s2', 2, 64, 96, 32, 32, 2, False),
             ('gather ragged', 1, 48, 80, 20, 28, 1, False), ('rgb', 2, 3, 64, 32, 32, 1, False), ('halo 16x16 tile', 1, 128, 128, 16, 16, 1, False)]
    for name, N, Cin, Cout, H, W, stride, up in cases:
        x = op_input(f'rp_{name}', (N, Cin, H, W))
        w = op_input(f'rpw_{name}', (Cout, Cin, 3, 3), 1.0 / (3.0 * Cin ** 0.5))
        b = op_input(f'rpb_{name}', (Cout,), 0.1)
        xin = F.interpolate(x, scale_factor=2, mode='nearest') if up else x
        ref = F.conv2d(F.pad(xin.double(), (1, 1, 1, 1), mode='reflect'), w.double(), b.double(), stride=stride)
        wp = w.permute(0, 2, 3, 1).contiguous().cuda()
        kw = dict(stride=stride, pad=1, ksize=3, upsample=up, reflect=True, mma=mma)
        if mma == L.MMA_X3 and Cin % 16 == 0:
            sc = ops.x3_scale_for(float(w.abs().max()))
            kw.update(wx3=ops.split_x3(wp.reshape(-1, Cin), sc).view(-1), x3_acc_scale=1.0 / sc)
        y = ops.conv(nhwc(x), wp, b.cuda(), **kw)
        err = (y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (name, err)


def test_reflect_is_refused_where_no_kernel_has_it():
    x = torch.zeros(1, 16, 16, 32, device='cuda')
    w = torch.zeros(32, 3, 3, 32, device='cuda')
    with pytest.raises(L.KeepHipError):
        ops.conv(x, w, None, reflect=True, mma=L.MMA_BF16, wb=w.to(torch.bfloat16))


@pytest.mark.parametrize("precision", ['x3', 'fp32'])
def test_parsenet128_vs_reference_golden_and_oracle(precision):
    W = PN.synth_parsenet_state_dict(seed=0, in_size=128, out_size=128)
    eng = PN.ParseNetEngine(W, in_size=128, out_size=128, precision=precision).to('cuda')
    x = op_input('parsenet128', (2, 3, 128, 128))
    logits = eng.logits(x.cuda()).cpu()
    assert logits.shape == (2, 19, 128, 128)
    grid = G['parsenet128_logit_grid']
    assert np.abs(logits[:, :, 1::4, 2::4].numpy() - grid).max() <= 3e-4 * np.abs(grid).max()
    with torch.no_grad():
        ref = FO.parsenet_forward(x, W, PN.parsenet_spec(in_size=128, out_size=128))
    err = (logits - ref).abs().max().item()
    print(f'ParseNet(128) [{precision}] max-abs logit diff vs the oracle: {err:.3e} (logit scale {ref.abs().max().item():.1f})')
    assert err <= 3e-4 * ref.abs().max().item()
    cls = eng.classes(nhwc(x)).cpu().numpy()
    safe = G['parsenet128_margin'].astype(np.float32) > 1e-2
    assert np.array_equal(cls[safe], G['parsenet128_classes'][safe])
    assert np.array_equal(cls, logits.argmax(1).numpy().astype(np.uint8))          # keep_channel_argmax == argmax of its logits


def test_parsenet512_batched_equals_one_by_one_and_reference_classes():
    """ParseNet(512, 512) as init_parsing_model builds it: the reference's classes for the golden face wherever its top-2
    margin exceeds 1e-2, and a batch of faces equal to the same faces one at a time (batch-invariant plans), through the
    drop-in ``face_parse(x)[0]`` call."""
    W = PN.synth_parsenet_state_dict(seed=0)
    eng = PN.ParseNetEngine(W).to('cuda')
    x = op_input('parsenet512', (1, 3, 512, 512))
    cls = eng.classes(nhwc(x)).cpu().numpy()
    safe = G['parsenet512_margin'].astype(np.float32) > 1e-2
    assert safe.mean() > 0.99 and np.array_equal(cls[safe], G['parsenet512_classes'][safe])
    fp = PN.EngineFaceParse(eng)
    xb = torch.cat([x, op_input('parsenet512b', (2, 3, 512, 512))], 0).cuda()
    out = fp(xb)[0]
    assert out.shape == (3, 19, 512, 512)
    grid = G['parsenet512_logit_grid']
    assert np.abs(out[:1, :, 3::16, 5::16].cpu().numpy() - grid).max() <= 3e-4 * np.abs(grid).max()
    for i in range(3):
        assert torch.equal(fp(xb[i:i + 1])[0][0], out[i])


@pytest.mark.parametrize("precision", ['x3', 'fp32'])
def test_retinaface_engine_vs_reference_golden(precision):
    """RetinaFace(resnet50) on the engine: raw head outputs against the golden composed from the reference's FPN / SSH / head
    modules (trunk: the restated torchvision ResNet-50, unpinned), then the whole detect_batch pipeline against the oracle
    network + the same host post-processing, frame by frame of a batch."""
    from comfyui_keep_amd.engine import retinaface as RF
    W = RF.synth_retinaface_state_dict(seed=0)
    eng = RF.RetinaFaceEngine(W, precision=precision).to('cuda')
    x = op_input('retinaface_img', (2, 3, 160, 224), 100.0)
    loc, cls, lm = (t.cpu() for t in eng.raw_outputs(nhwc(x)))
    conf = torch.softmax(cls, -1)
    for got, key, tol in ((loc, 'retinaface_loc', 3e-4), (conf, 'retinaface_conf', 3e-4), (lm, 'retinaface_landm', 3e-4)):
        err = np.abs(got.numpy() - G[key]).max()
        print(f'RetinaFace [{precision}] {key}: max-abs diff {err:.3e} (scale {np.abs(G[key]).max():.1f})')
        assert err <= tol * max(1.0, np.abs(G[key]).max()), (key, err)
    # full pipeline on uint8 frames of a ragged size (stride-32 maps 5 x 7 -> ceil), batch of 3 == one by one == oracle
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (3, 150, 210, 3), generator=g, dtype=torch.uint8)
    dets = eng.detect_batch(frames, conf_threshold=0.8)
    assert len(dets) == 3 and all(d.ndim == 2 and d.shape[1] == 15 for d in dets)
    single = RF.EngineRetinaFace(eng).detect_faces(frames[1].numpy(), 0.8)
    assert np.array_equal(single, dets[1])
    xf = frames.float().permute(0, 3, 1, 2) - torch.tensor(RF.MEAN_BGR).view(1, 3, 1, 1)
    with torch.no_grad():
        rloc, rconf, rlm = FO.retinaface_forward(xf, W)
    pri = RF.prior_boxes(150, 210)
    for i in range(3):
        sc = rconf[i, :, 1].numpy()
        sure = (np.abs(sc - 0.8) > 1e-3)                       # anchors whose score is not within rounding of the threshold
        boxes = RF.decode_boxes(rloc[i].numpy(), pri, RF.CFG_RE50['variance']) * np.array([210, 150, 210, 150], np.float32)
        keep_ref = np.where((sc > 0.8) & sure)[0]
        assert len(dets[i]) > 0
        # every engine detection is a reference candidate (same box within 1e-2 px, same score within 1e-4)
        for d in dets[i]:
            j = np.argmin(np.abs(boxes - d[:4]).sum(1))
            assert np.abs(boxes[j] - d[:4]).max() <= 1e-2 and abs(sc[j] - d[4]) <= 1e-4, (i, d[:5], boxes[j], sc[j])
        assert len(dets[i]) <= len(keep_ref) + int((~sure).sum())


@pytest.mark.parametrize("precision", ['x3', 'fp32'])
def test_retinaface_mobile025_engine_vs_reference_golden(precision):
    """RetinaFace('mobile0.25') on the engine against the golden of the reference's OWN modules from image to heads (MobileNetV1,
    FPN, SSH, heads: nothing restated in between), then detect_batch against the oracle network + host post-processing."""
    from comfyui_keep_amd.engine import retinaface as RF
    W = RF.synth_retinaface_state_dict(seed=0, backbone='mobile0.25')
    eng = RF.RetinaFaceEngine(W, precision=precision).to('cuda')
    assert eng.backbone == 'mobile0.25'
    x = op_input('retinaface_mnet_img', (2, 3, 160, 224), 100.0)
    loc, cls, lm = (t.cpu() for t in eng.raw_outputs(nhwc(x)))
    conf = torch.softmax(cls, -1)
    for got, key, tol in ((loc, 'mnet_loc', 3e-4), (conf, 'mnet_conf', 3e-4), (lm, 'mnet_landm', 3e-4)):
        err = np.abs(got.numpy() - G[key]).max()
        print(f'RetinaFace mobile0.25 [{precision}] {key}: max-abs diff {err:.3e} (scale {np.abs(G[key]).max():.1f})')
        assert err <= tol * max(1.0, np.abs(G[key]).max()), (key, err)
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (3, 150, 210, 3), generator=g, dtype=torch.uint8)
    dets = eng.detect_batch(frames, conf_threshold=0.7)
    single = RF.EngineRetinaFace(eng).detect_faces(frames[2].numpy(), 0.7)
    assert np.array_equal(single, dets[2])
    xf = frames.float().permute(0, 3, 1, 2) - torch.tensor(RF.MEAN_BGR).view(1, 3, 1, 1)
    with torch.no_grad():
        rloc, rconf, rlm = FO.retinaface_forward(xf, W, backbone='mobile0.25')
    pri = RF.prior_boxes(150, 210, RF.CFG_MNET)
    total = 0
    for i in range(3):
        sc = rconf[i, :, 1].numpy()
        sure = (np.abs(sc - 0.7) > 1e-3)
        boxes = RF.decode_boxes(rloc[i].numpy(), pri, RF.CFG_MNET['variance']) * np.array([210, 150, 210, 150], np.float32)
        lms = RF.decode_landmarks(rlm[i].numpy(), pri, RF.CFG_MNET['variance']) * np.array([210, 150] * 5, np.float32)
        keep_ref = np.where((sc > 0.7) & sure)[0]
        total += len(dets[i])
        for d in dets[i]:
            j = np.argmin(np.abs(boxes - d[:4]).sum(1))
            assert np.abs(boxes[j] - d[:4]).max() <= 1e-2 and abs(sc[j] - d[4]) <= 1e-4 and np.abs(lms[j] - d[5:]).max() <= 1e-2
        assert len(dets[i]) <= len(keep_ref) + int((~sure).sum())
    assert total > 0


def test_dwconv3x3_vs_torch_grouped_convolution():
    """keep_dwconv3x3 (depthwise 3x3, padding 1, stride 1 | 2, bias + activation) against F.conv2d(groups = C) on odd and even map
    sizes, every activation the trunk uses; float32 FMA chains of 9 taps: 1e-6 relative."""
    import torch.nn.functional as F
    for (N, H, W, C, stride, act) in ((2, 17, 23, 8, 1, L.ACT_LRELU01), (1, 40, 56, 64, 2, L.ACT_LRELU01), (3, 9, 9, 256, 2, L.ACT_NONE),
                                      (1, 16, 16, 32, 1, L.ACT_RELU)):
        x = op_input(f'dw_x{C}', (N, C, H, W))
        w = op_input(f'dw_w{C}', (C, 1, 3, 3), 0.5)
        b = op_input(f'dw_b{C}', (C,), 0.2)
        ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1, groups=C)
        ref = {L.ACT_LRELU01: lambda t: F.leaky_relu(t, 0.1), L.ACT_RELU: F.relu, L.ACT_NONE: lambda t: t}[act](ref)
        xd, wd, bd = nhwc(x), w[:, 0].permute(1, 2, 0).contiguous().cuda(), b.cuda()
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = torch.empty((N, Ho, Wo, C), device='cuda')
        L.call('keep_dwconv3x3', xd, wd, bd, out, N, H, W, C, stride, act)
        got = out.cpu().permute(0, 3, 1, 2).double()
        assert got.shape == ref.shape and (got - ref).abs().max() <= 2e-6 * ref.abs().max(), (C, stride, (got - ref).abs().max())
    with pytest.raises(L.KeepHipError):
        L.call('keep_dwconv3x3', xd, wd, bd, out, N, H, W, 6, 1, L.ACT_NONE)          # C % 4 != 0: loud


@pytest.mark.parametrize("name", ['YOLOv5n', 'YOLOv5l'])
@pytest.mark.parametrize("precision", ['x3', 'fp32'])
def test_yolov5_face_engine_vs_reference_golden(name, precision):
    """YOLOv5n / YOLOv5l on the engine (StemBlock, ShuffleV2 / C3 / SPP blocks, Detect decode on the device) against the output of
    the reference's own Model(yaml) on the same synthetic weights; batched == one by one; through the drop-in for
    ``YoloDetector.detector`` (NCHW in, ``(pred, None)`` out)."""
    from comfyui_keep_amd.engine import yoloface as YF
    W = YF.synth_yolo_state_dict(name, seed=0)
    eng = YF.YoloFaceEngine(W, precision=precision).to('cuda')
    x = op_input(f'yolo_img_{name}', (2, 3, 96, 128)).mul(0.5).add(0.5).clamp(0, 1)
    pred = eng.forward_nhwc(nhwc(x)).cpu().numpy()
    ref = G[f'{name.lower()}_pred']
    assert pred.shape == ref.shape
    err_px = np.abs(pred[..., :4] - ref[..., :4]).max()
    err_lm = np.abs(pred[..., 5:15] - ref[..., 5:15]).max()
    err_sc = np.abs(pred[..., [4, 15]] - ref[..., [4, 15]]).max()
    print(f'{name} [{precision}]: box {err_px:.2e} px (max {np.abs(ref[..., :4]).max():.0f}), landmarks {err_lm:.2e} (max {np.abs(ref[..., 5:15]).max():.0f}), scores {err_sc:.2e}')
    assert err_px <= 2e-3 and err_sc <= 1e-5 and err_lm <= 3e-5 * np.abs(ref[..., 5:15]).max()
    model = YF.EngineYoloModel(eng)
    out = model(x.cuda())
    assert out[1] is None and np.array_equal(out[0].cpu().numpy(), pred)
    one = eng.forward_nhwc(nhwc(x[1:2])).cpu().numpy()
    assert np.abs(one[0] - pred[1]).max() <= 1e-4 * max(1.0, np.abs(pred).max())
    with pytest.raises(ValueError):
        eng.forward_nhwc(torch.zeros(1, 100, 128, 3, device='cuda'))


def test_yolo_helper_kernels_vs_torch():
    """keep_maxpool2d (ceil-mode 2x2 stride 2, k x k stride 1 on channel slices), keep_slice_copy (concat, nearest x2 + concat),
    keep_channel_shuffle2 and keep_yolo_decode against their torch statements: all bit-exact except the decode's sigmoid."""
    x = op_input('yk_x', (2, 24, 17, 21))
    xd = nhwc(x)
    for (k, s, p, ceil) in ((2, 2, 0, True), (3, 1, 1, False), (5, 1, 2, False), (7, 1, 3, False)):
        ref = F.max_pool2d(x, k, s, p, ceil_mode=ceil)
        Ho, Wo = ref.shape[2:]
        wide = torch.full((2, Ho, Wo, 40), -7.0, device='cuda')
        L.call('keep_maxpool2d', xd.view(-1)[8:], wide.view(-1)[12:], 2, 17, 21, 16, 24, 40, k, s, p, Ho, Wo)      # channels 8..24 -> 12..28
        assert torch.equal(wide[..., 12:28].cpu(), ref[:, 8:24].permute(0, 2, 3, 1)) and float(wide[..., :12].max()) == -7.0 == float(wide[..., 28:].min())
    a, b = op_input('yk_a', (2, 8, 6, 10)), op_input('yk_b', (2, 12, 12, 20))
    cat = torch.zeros((2, 12, 20, 20), device='cuda')
    L.call('keep_slice_copy', nhwc(a), cat.view(-1), 2, 12, 20, 8, 8, 20, 1)
    L.call('keep_slice_copy', nhwc(b), cat.view(-1)[8:], 2, 12, 20, 12, 12, 20, 0)
    ref = torch.cat((F.interpolate(a, scale_factor=2, mode='nearest'), b), 1)
    assert torch.equal(cat.cpu().permute(0, 3, 1, 2), ref)
    src = op_input('yk_s', (1, 32, 5, 7))
    y = op_input('yk_y', (1, 16, 5, 7))
    out = torch.empty((1, 5, 7, 32), device='cuda')
    L.call('keep_channel_shuffle2', nhwc(src), nhwc(y), out, 35, 16, 32, 16)
    t = torch.cat((src[:, :16], y), 1)
    ref = t.view(1, 2, 16, 5, 7).transpose(1, 2).contiguous().view(1, 32, 5, 7)
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), ref)
    raw = op_input('yk_raw', (2, 48, 4, 6), 2.0)
    pred = torch.zeros((2, 100 + 72, 16), device='cuda')
    anchors = torch.tensor([23., 29., 43., 55., 73., 105.], device='cuda')
    L.call('keep_yolo_decode', nhwc(raw), pred, 2, 4, 6, 16.0, anchors, 100, 172)
    r = raw.view(2, 3, 16, 4, 6).permute(0, 1, 3, 4, 2)
    yv, xv = torch.meshgrid(torch.arange(4), torch.arange(6), indexing='ij')
    grid = torch.stack((xv, yv), 2).view(1, 1, 4, 6, 2).float()
    ag = anchors.cpu().view(1, 3, 1, 1, 2)
    o = torch.zeros_like(r)
    o[..., [0, 1, 2, 3, 4, 15]] = r[..., [0, 1, 2, 3, 4, 15]].sigmoid()
    o[..., 0:2] = (o[..., 0:2] * 2.0 - 0.5 + grid) * 16.0
    o[..., 2:4] = (o[..., 2:4] * 2) ** 2 * ag
    for q in range(5):
        o[..., 5 + 2 * q:7 + 2 * q] = r[..., 5 + 2 * q:7 + 2 * q] * ag + grid * 16.0
    got = pred.cpu()
    assert float(got[:, :100].abs().max()) == 0.0 and (got[:, 100:] - o.reshape(2, 72, 16)).abs().max() <= 2e-5 * o.abs().max()


def tie_canon(d):
    """Rows of one frame's detections with runs of EQUAL scores put in a canonical order (by x1, y1): `scores.argsort()[::-1]`
    (retinaface.py:240) leaves the order inside such a run to numpy's introsort; the device orders it by descending anchor index
    (what numpy gives whenever its sort is stable: up to 16 candidates)."""
    return d[np.lexsort((d[:, 1], d[:, 0], -d[:, 4]))]


def test_retina_decode_on_the_device_equals_the_host_decoder():
    """keep_retina_decode (scores, threshold, decode / decode_landm on the device; only survivors cross PCIe) against the numpy
    decoder of rounds 2-3 on the SAME head rows: the same anchors survive (up to scores within 1e-6 of the threshold), boxes /
    landmarks agree to 1e-4 px on a 640-px frame, the final detections (after NMS) are the same set in the same order; a frame with
    more survivors than the compact list holds takes the host decoder; float frames (16-bit sources) take the float path."""
    from comfyui_keep_amd.engine import retinaface as RF
    eng = RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0)).to('cuda')
    g = torch.Generator().manual_seed(9)
    frames = torch.randint(0, 256, (3, 320, 448, 3), generator=g, dtype=torch.uint8)
    H, W = 320, 448
    pri = RF.prior_boxes(H, W)
    scale, scale1 = np.array([W, H, W, H], np.float32), np.array([W, H] * 5, np.float32)
    x = frames.cuda().float() - torch.tensor(RF.MEAN_BGR, device='cuda')
    heads = eng.raw_heads(x).float().cpu().numpy()
    for thr in (0.9, 0.6):
        got = eng.detect_batch(frames, thr)
        for i in range(3):
            ref = eng._host_decode(heads[i], pri, scale, scale1, thr, 0.4)
            assert len(ref) > 3 and got[i].shape == ref.shape, (thr, i, got[i].shape, ref.shape)
            assert np.array_equal(got[i][:, 4], ref[:, 4]) or np.abs(got[i][:, 4] - ref[:, 4]).max() <= 1e-6      # the same descending score sequence
            a, b = tie_canon(got[i]), tie_canon(ref)
            assert np.abs(a - b).max() <= 1e-4 * max(H, W), (thr, i, np.abs(a - b).max())
    # keep_retina_nms (ordering + greedy IoU suppression on the device) against the numpy ordering / NMS on the same survivors: the same
    # rows in the same order, bit for bit (same float32 IoU arithmetic; no two survivors of these frames share a score)
    assert eng.device_nms
    on_device = eng.detect_batch(frames, 0.6)
    eng.device_nms = False
    on_host = eng.detect_batch(frames, 0.6)
    eng.device_nms = True
    ties = 0
    for a, b in zip(on_device, on_host):
        assert a.shape == b.shape and len(a) > 3 and np.array_equal(tie_canon(a), tie_canon(b))
        ties += int(not np.array_equal(a, b))
    # round 5: a frame in which two survivors share a score bit pattern is handed back to the host decoder (numpy's own
    # introsort order applies there), so device and host results are the same rows in the same order on EVERY frame
    assert ties == 0, f'{ties} of 3 frames differ from the numpy order'
    eng.max_survivors = 8                                   # overflow of the compact list: the frame is decoded on the host
    few = eng.detect_batch(frames, 0.6)
    eng.max_survivors = 4096
    full = eng.detect_batch(frames, 0.6)
    for a, b in zip(few, full):
        assert a.shape == b.shape and np.abs(tie_canon(a) - tie_canon(b)).max() <= 1e-4 * max(H, W)
    as_float = eng.detect_batch(frames.double().numpy(), 0.9)                     # read_image's float64 frames
    for a, b in zip(as_float, eng.detect_batch(frames, 0.9)):
        assert np.array_equal(a, b)


def test_retina_tied_frames_are_ordered_on_the_host_and_suppressed_on_the_device():
    """Frames whose survivors share score bit patterns (keep_retina_nms -> -2): ``_tied_frames`` (host: the reference's ordering calls on
    scores + anchor indices; device: keep_retina_nms_ordered) returns what the all-numpy path (``_host_order_nms``) returns, bit for bit --
    crafted frames with long runs of equal scores inside overlapping clusters (the order of a run decides which box survives), arrival
    order shuffled, 30 .. 3000 survivors; and through ``detect_batch`` on mobile0.25 with synthetic weights (its softmax saturates at 1.0f)."""
    from comfyui_keep_amd.engine import retinaface as RF
    eng = RF.RetinaFaceEngine(RF.synth_retinaface_state_dict(seed=0, backbone='mobile0.25')).to('cuda')
    rng = np.random.default_rng(5)
    cap = 4096
    sizes = [30, 700, 3000, 17]
    dets = np.zeros((len(sizes), cap, 16), np.float32)
    for f, n in enumerate(sizes):
        cx, cy = rng.uniform(50, 600, n // 6 + 1), rng.uniform(50, 400, n // 6 + 1)
        which = rng.integers(0, len(cx), n)
        w, h = rng.uniform(30, 60, n), rng.uniform(30, 60, n)
        x1, y1 = cx[which] + rng.normal(0, 6, n) - w / 2, cy[which] + rng.normal(0, 6, n) - h / 2
        dets[f, :n, 0:4] = np.stack((x1, y1, x1 + w, y1 + h), 1)
        dets[f, :n, 4] = rng.choice(np.array([1.0, 0.99999994, 0.9999, 0.98, 0.975], np.float32), n)        # long runs of equal scores
        dets[f, :n, 5:15] = rng.uniform(0, 640, (n, 10))
        dets[f, :n, 15] = rng.permutation(20000)[:n]                                                     # anchor indices, arrival order shuffled
    d = torch.from_numpy(dets).cuda()
    counts = torch.tensor(sizes, dtype=torch.int32, device='cuda')
    kept, kcnt = torch.empty_like(d), torch.empty(len(sizes), dtype=torch.int32, device='cuda')
    L.call('keep_retina_nms', d, counts, kept, kcnt, len(sizes), cap, 0.4)
    assert kcnt.cpu().tolist() == [-2] * len(sizes)
    got = eng._tied_frames(d, counts, list(range(len(sizes))), cap, 0.4)
    for f, n in enumerate(sizes):
        ref = eng._host_order_nms(dets[f, :n], 0.4)
        assert got[f].shape == ref.shape and 3 < len(ref) < n and np.array_equal(got[f], ref), (f, got[f].shape, ref.shape)
    part = eng._tied_frames(d, counts, [2, 0], cap, 0.4)                 # a subset of the chunk's frames, in the caller's order
    assert np.array_equal(part[0], got[2]) and np.array_equal(part[1], got[0])
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (3, 320, 448, 3), generator=g, dtype=torch.uint8)
    on_device = eng.detect_batch(frames, 0.97)
    os.environ['KEEP_AMD_DEVICE_TIED_NMS'] = '0'
    try:
        on_host = eng.detect_batch(frames, 0.97)
    finally:
        del os.environ['KEEP_AMD_DEVICE_TIED_NMS']
    for a, b in zip(on_device, on_host):
        assert len(a) > 3 and np.array_equal(a, b)


def test_yolo_letterbox_kernel_vs_oracle():
    """keep_yolo_letterbox_u8 (BGR2RGB + cv2.resize(INTER_LINEAR) + 114 border + / 255, NHWC) bit-exact against the oracle's restatement
    of ``_preprocess`` and against the golden network inputs of the reference run (tests/golden/yolo_prepost.npz), on the golden frames
    and on product-like sizes: 640 x 1137 (a 720p frame after the helper's resize: scaled by 1152 / 1137, padded to 704 x 1152), a
    padding-only size, a tall frame (left / right padding), strong magnification."""
    from comfyui_keep_amd.engine import yoloface as YF
    g = np.load(os.path.join(GOLDEN, 'yolo_prepost.npz'))
    cases = [(tag, frames) for tag, (frames, _) in FO.yolo_prepost_inputs().items()]
    rng = np.random.default_rng(7)
    for hw in ((640, 1137), (352, 640), (517, 333), (23, 41), (300, 1000)):
        cases.append((None, rng.integers(0, 256, (2, *hw, 3), dtype=np.uint8)))
    smooth = (np.add.outer(np.arange(97), np.arange(131))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8)[None]
    cases.append((None, smooth))
    for tag, frames in cases:
        N, H, W, _ = frames.shape
        (rh, rw), (top, left), (H2, W2) = YF.letterbox_geometry(H, W)
        out = torch.empty((N, H2, W2, 3), device='cuda')
        L.call('keep_yolo_letterbox_u8', torch.from_numpy(frames).cuda(), out, N, H, W, rh, rw, top, left, H2, W2, 1)
        ref = FO.yolo_preprocess(list(frames))
        got = out.cpu().permute(0, 3, 1, 2)
        assert torch.equal(got, ref), (tag, H, W, float((got - ref).abs().max()) * 255)
        if tag is not None:
            for n in range(N):
                assert np.array_equal(got[n].numpy(), g[f'{tag}_x{n}'].astype(np.float32) / np.float32(255.0))
    # swap_rb = 0 keeps the channel order; bad geometry and the 2x reduction (cv2's INTER_AREA path) are refused
    f = torch.from_numpy(cases[0][1]).cuda()
    out = torch.empty((2, 160, 160, 3), device='cuda')
    L.call('keep_yolo_letterbox_u8', f, out, 2, 100, 160, 100, 160, 30, 0, 160, 160, 0)
    assert torch.equal(out[:, 30:130].cpu(), torch.from_numpy(cases[0][1]).float() / 255.0)
    with pytest.raises(L.KeepHipError):
        L.call('keep_yolo_letterbox_u8', f, out, 2, 100, 160, 100, 160, 70, 0, 160, 160, 1)
    with pytest.raises(L.KeepHipError):
        L.call('keep_yolo_letterbox_u8', f, out, 2, 100, 160, 50, 80, 0, 0, 160, 160, 1)


def _yolo_stub_detector(name='YOLOv5n', pred=None):
    """A YoloDetector as keep_model_loader leaves it (detector = EngineYoloModel, target_size None, min_face 10), without the reference
    class: the device path reads exactly these attributes.  ``pred``: replace the network's output (crafted predictions)."""
    import types
    from comfyui_keep_amd.engine import yoloface as YF
    eng = YF.YoloFaceEngine(YF.synth_yolo_state_dict(name, seed=0)).to('cuda')
    if pred is not None:
        real = eng.forward_nhwc
        state = {'i': 0}

        def forward(x):
            shape = real(x).shape                      # (the network still runs: shapes are checked against it)
            out = torch.from_numpy(pred[state['i']:state['i'] + x.shape[0]]).cuda()
            state['i'] += x.shape[0]
            assert tuple(out.shape) == tuple(shape)
            return out
        eng.forward_nhwc = forward
    return types.SimpleNamespace(detector=YF.EngineYoloModel(eng), target_size=None, min_face=10, device='cuda')


def test_yolo_detect_batch_device_vs_reference_golden():
    """``yolo_detect_batch`` on the device (letterbox kernel -> network -> keep_yolo_select -> keep_retina_nms -> host tail on the kept
    rows) against what the reference's own ``detect_faces`` returned for the golden frames and crafted predictions, at its default
    thresholds and at the helper's 0.97; a frame without faces is None; chunks smaller than the batch; the overflow (cap 8) and
    equal-conf hand-backs finish on the host with the same result."""
    from comfyui_keep_amd.engine import yoloface as YF
    g = np.load(os.path.join(GOLDEN, 'yolo_prepost.npz'))
    for tag, (frames, pred) in FO.yolo_prepost_inputs().items():
        for conf, name in ((0.7, 'default'), (0.97, 'helper')):
            want = [g[f'{tag}_{name}_det{n}'] for n in range(2)]
            for kw in ({}, {'max_frames': 1}, {'cap': 8}):
                det = _yolo_stub_detector(pred=pred)
                got = (YF.yolo_detect_batch(det, frames, conf, 0.5) if not kw else YF.yolo_detect_batch_device(det, frames, conf, 0.5, **kw))
                for n in range(2):
                    if len(want[n]) == 0:
                        assert got[n] is None, (tag, name, n, kw)
                    else:
                        assert got[n].dtype == np.int64 and np.array_equal(got[n], want[n]), (tag, name, n, kw, got[n], want[n])
    # two candidates with the same conf: the device hands the frame back (-2) and the host orders them (stable: first row first)
    frames, pred = FO.yolo_prepost_inputs()['pad']
    pred = pred.copy()
    near = (np.abs(pred[0, :, 0] - 48.0) < 10) & (np.abs(pred[0, :, 1] - 80.0) < 10) & (pred[0, :, 4] > 0.6)      # the cluster around (48, 80)
    rows = np.flatnonzero(near)[:2]
    pred[0, rows, 4] = pred[0, rows, 15] = 0.9990234375            # the two best of the frame, the same conf, overlapping, different boxes /
    assert len(rows) == 2 and not np.array_equal(pred[0, rows[0], :4], pred[0, rows[1], :4])      # landmarks: the order decides which one survives
    det = _yolo_stub_detector(pred=pred)
    got = YF.yolo_detect_batch(det, frames, 0.7, 0.5)
    ref = FO.yolo_postprocess(pred[0], (160, 160), (100, 160), 0.7, 0.5)
    assert np.array_equal(got[0], ref)


def test_yolo_detect_batch_device_on_the_network_vs_oracle_postprocess():
    """The whole device path on the engine's own predictions (synthetic weights: objectness spans its range, hundreds of candidates
    at a low threshold): equal to the oracle's post-processing of the same prediction tensor, frame by frame; selection kernel counts =
    the number of rows above both thresholds."""
    from comfyui_keep_amd.engine import yoloface as YF
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (3, 176, 301, 3), dtype=np.uint8)
    det = _yolo_stub_detector('YOLOv5n')
    (rh, rw), (top, left), (H2, W2) = YF.letterbox_geometry(176, 301)
    x = torch.empty((3, H2, W2, 3), device='cuda')
    L.call('keep_yolo_letterbox_u8', torch.from_numpy(frames).cuda(), x, 3, 176, 301, rh, rw, top, left, H2, W2, 1)
    pred = det.detector.engine.forward_nhwc(x)
    p = pred.cpu().numpy()
    for conf in (0.5, 0.3):
        n_cand = [(int(((p[i, :, 4] > np.float32(conf)) & (p[i, :, 15] * p[i, :, 4] > np.float32(conf))).sum())) for i in range(3)]
        dets = torch.empty((3, 4096, 16), device='cuda')
        counts = torch.zeros(3, dtype=torch.int32, device='cuda')
        L.call('keep_yolo_select', pred, dets, counts, 3, p.shape[1], 4096, conf)
        assert counts.cpu().tolist() == n_cand
        got = YF.yolo_detect_batch_device(det, frames, conf, 0.5, cap=4096)
        print(f'yolo device path: conf {conf}: candidates {n_cand}, faces {[0 if r is None else len(r) for r in got]}')
        for i in range(3):
            ref = FO.yolo_postprocess(p[i], (H2, W2), (176, 301), conf, 0.5)
            assert (got[i] is None and ref is None) or np.array_equal(got[i], ref), (conf, i)
