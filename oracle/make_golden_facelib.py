"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the face-analysis networks from the IMPORTED REFERENCE
modules under /root/reference (build container only) on deterministic synthetic weights / inputs:

    python oracle/make_golden_facelib.py        # a few seconds of CPU

  tests/golden/facelib.npz
    parsenet128_*        reference ParseNet(in_size=128, out_size=128) on op_input('parsenet128', (2,3,128,128)): arg-max
                         classes, top-2 margins, a strided logit digest
    parsenet512_classes  reference ParseNet(512, 512) arg-max classes (uint8) + a strided logit digest on one 512x512 input
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import parsenet as PN  # noqa: E402
from make_golden import op_input  # noqa: E402

REF = os.environ.get('KEEP_REFERENCE_ROOT', '/root/reference')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def retinaface_golden():
    """RetinaFace(resnet50): the reference's OWN FPN / SSH / head modules (retinaface_net.py) and prior / decode functions
    (retinaface_utils.py, imported with a stub `torchvision` whose only touched member is ops.nms) on synthetic weights; the
    ResNet-50 trunk (torchvision in the reference, absent here) is the oracle's restatement -- the trunk is NOT pinned."""
    import types
    import facelib_oracle as FO
    from comfyui_keep_amd.engine import retinaface as RF
    tv = types.ModuleType('torchvision')
    tv.ops = types.SimpleNamespace(nms=lambda boxes, scores, iou_threshold: torch.tensor(RF.nms(
        np.concatenate([boxes.numpy(), scores.numpy()[:, None]], 1), iou_threshold), dtype=torch.int64))
    sys.modules.setdefault('torchvision', tv)
    base = os.path.join(REF, 'modules', 'deps', 'wm_facelib', 'detection', 'retinaface')
    net = _load(os.path.join(base, 'retinaface_net.py'), 'ref_retinaface_net')
    utils = _load(os.path.join(base, 'retinaface_utils.py'), 'ref_retinaface_utils')
    W = RF.synth_retinaface_state_dict(seed=0)
    x = op_input('retinaface_img', (2, 3, 160, 224), 100.0)            # BGR minus mean, O(100) like real frames
    with torch.no_grad():
        feats = FO.resnet50_trunk(x, W)
        fpn = net.FPN([512, 1024, 2048], 256).eval()
        fpn.load_state_dict({k[4:]: v for k, v in W.items() if k.startswith('fpn.')}, strict=True)
        pyr = fpn(feats)
        sshs = []
        for k in (1, 2, 3):
            m = net.SSH(256, 256).eval()
            m.load_state_dict({kk[5:]: v for kk, v in W.items() if kk.startswith(f'ssh{k}.')}, strict=True)
            sshs.append(m(pyr[k - 1]))
        heads = {}
        for name, maker in (('ClassHead', net.make_class_head), ('BboxHead', net.make_bbox_head), ('LandmarkHead', net.make_landmark_head)):
            hs = maker(fpn_num=3, inchannels=256).eval()
            hs.load_state_dict({kk[len(name) + 1:]: v for kk, v in W.items() if kk.startswith(name + '.')}, strict=True)
            heads[name] = torch.cat([hs[i](f) for i, f in enumerate(sshs)], dim=1)
    conf = torch.softmax(heads['ClassHead'], dim=-1)
    cfg = dict(RF.CFG_RE50)
    priors = utils.PriorBox(cfg, image_size=(160, 224)).forward()
    boxes = utils.decode(heads['BboxHead'][0], priors, cfg['variance'])
    lms = utils.decode_landm(heads['LandmarkHead'][0], priors, cfg['variance'])
    print('RetinaFace golden: conf range', float(conf[..., 1].min()), float(conf[..., 1].max()), 'priors', tuple(priors.shape))
    extra = retinaface_mnet_golden(net, utils)
    return {**extra, 'retinaface_loc': heads['BboxHead'].numpy().astype(np.float32), 'retinaface_conf': conf.numpy().astype(np.float32),
            'retinaface_landm': heads['LandmarkHead'].numpy().astype(np.float32), 'retinaface_priors': priors.numpy().astype(np.float32),
            'retinaface_boxes0': boxes.numpy().astype(np.float32), 'retinaface_lms0': lms.numpy().astype(np.float32)}


def retinaface_mnet_golden(net, utils):
    """RetinaFace('mobile0.25') END TO END on the reference's own modules: MobileNetV1 (stage1 / stage2 / stage3, what
    IntermediateLayerGetter returns for cfg_mnet's return_layers, retinaface.py:37-41,96-98), FPN([64, 128, 256], 64), three
    SSH(64, 64) and the heads (retinaface.py:100-122), composed in the order of RetinaFace.forward (retinaface.py:129-146).
    (RetinaFace itself is not instantiated: its module imports torchvision and cv2.)"""
    from comfyui_keep_amd.engine import retinaface as RF
    W = RF.synth_retinaface_state_dict(seed=0, backbone='mobile0.25')
    x = op_input('retinaface_mnet_img', (2, 3, 160, 224), 100.0)
    with torch.no_grad():
        body = net.MobileNetV1().eval()
        missing = body.load_state_dict({k[5:]: v for k, v in W.items() if k.startswith('body.')}, strict=False)
        assert not missing.unexpected_keys and set(missing.missing_keys) == {'fc.weight', 'fc.bias'}, missing     # (the classifier is not part of the detector)
        feats = []
        h = x
        for stage in (body.stage1, body.stage2, body.stage3):
            h = stage(h)
            feats.append(h)
        fpn = net.FPN([64, 128, 256], 64).eval()
        fpn.load_state_dict({k[4:]: v for k, v in W.items() if k.startswith('fpn.')}, strict=True)
        pyr = fpn(feats)
        sshs = []
        for k in (1, 2, 3):
            m = net.SSH(64, 64).eval()
            m.load_state_dict({kk[5:]: v for kk, v in W.items() if kk.startswith(f'ssh{k}.')}, strict=True)
            sshs.append(m(pyr[k - 1]))
        heads = {}
        for name, maker in (('ClassHead', net.make_class_head), ('BboxHead', net.make_bbox_head), ('LandmarkHead', net.make_landmark_head)):
            hs = maker(fpn_num=3, inchannels=64).eval()
            hs.load_state_dict({kk[len(name) + 1:]: v for kk, v in W.items() if kk.startswith(name + '.')}, strict=True)
            heads[name] = torch.cat([hs[i](f) for i, f in enumerate(sshs)], dim=1)
    conf = torch.softmax(heads['ClassHead'], dim=-1)
    cfg = dict(RF.CFG_MNET)
    priors = utils.PriorBox(cfg, image_size=(160, 224)).forward()
    boxes = utils.decode(heads['BboxHead'][0], priors, cfg['variance'])
    lms = utils.decode_landm(heads['LandmarkHead'][0], priors, cfg['variance'])
    print('RetinaFace(mobile0.25) golden: conf range', float(conf[..., 1].min()), float(conf[..., 1].max()),
          'feature rms', [float(f.pow(2).mean().sqrt()) for f in feats])
    return {'mnet_stage_grid': torch.cat([f[:, ::8, ::3, ::5].reshape(-1) for f in feats]).numpy().astype(np.float32),
            'mnet_loc': heads['BboxHead'].numpy().astype(np.float32), 'mnet_conf': conf.numpy().astype(np.float32),
            'mnet_landm': heads['LandmarkHead'].numpy().astype(np.float32),
            'mnet_boxes0': boxes.numpy().astype(np.float32), 'mnet_lms0': lms.numpy().astype(np.float32)}


def yolo_golden():
    """YOLOv5n / YOLOv5l face detectors: the reference's OWN ``Model(cfg=yolov5{n,l}.yaml)`` (yolov5face/models/yolo.py: parse_model
    of its yaml, every block of common.py, Detect) on synthetic weights, image -> ``model(x)[0]``.  The package ``__init__`` files
    are not run (namespace packages with the real paths); cv2 and torchvision are imported by utils/datasets.py / general.py at
    module level only (letterbox / NMS: not on this path) and are stubbed empty."""
    import types
    from comfyui_keep_amd.engine import yoloface as YF
    deps = os.path.join(REF, 'modules', 'deps')
    for name, sub in (('wm_facelib', ''), ('wm_facelib.detection', 'detection'), ('wm_facelib.detection.yolov5face', 'detection/yolov5face'),
                      ('wm_facelib.detection.yolov5face.models', 'detection/yolov5face/models'),
                      ('wm_facelib.detection.yolov5face.utils', 'detection/yolov5face/utils')):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(deps, 'wm_facelib', sub)]
            sys.modules[name] = m
    for stub in ('cv2', 'torchvision'):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    import importlib
    yolo = importlib.import_module('wm_facelib.detection.yolov5face.models.yolo')
    out = {}
    for name, fn in (('YOLOv5n', 'yolov5n.yaml'), ('YOLOv5l', 'yolov5l.yaml')):
        model = yolo.Model(cfg=os.path.join(deps, 'wm_facelib', 'detection', 'yolov5face', 'models', fn)).eval()
        W = YF.synth_yolo_state_dict(name, seed=0)
        ref_spec = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert ref_spec == {k: tuple(v) for k, v in YF.yolo_state_dict_spec(name).items()}, (name, 'spec drift',
                                                                                             set(ref_spec) ^ set(YF.yolo_state_dict_spec(name)))
        assert [float(s) for s in model.stride] == list(YF.STRIDES)
        model.load_state_dict(W, strict=True)
        x = op_input(f'yolo_img_{name}', (2, 3, 96, 128)).mul(0.5).add(0.5).clamp(0, 1)      # [0, 1] like _preprocess's uint8 / 255
        with torch.no_grad():
            pred = model(x)[0]
        key = name.lower()
        out[f'{key}_pred'] = pred.numpy().astype(np.float32)
        print(f'{name} golden: pred', tuple(pred.shape), 'obj range', float(pred[..., 4].min()), float(pred[..., 4].max()),
              'xy range', float(pred[..., :2].min()), float(pred[..., :2].max()), '|raw landmarks| max', float(pred[..., 5:15].abs().max()))
    return out


def yolo_prepost_golden():
    """The reference's OWN ``YoloDetector.detect_faces`` (yolov5face/face_detector.py:113-141: BGR2RGB, _preprocess with letterbox, the
    network call, _postprocess with non_max_suppression_face / scale_coords / scale_coords_landmarks, the assembly) run frame by frame on
    ``facelib_oracle.yolo_prepost_inputs``; its network is a recorder that keeps the pre-processed tensor and returns the crafted prediction.  Absent
    dependencies: ``cv2`` = {cvtColor(BGR2RGB): channel flip, copyMakeBorder(constant): np.pad, resize(INTER_LINEAR):
    facelib_oracle.cv2_resize_linear_u8 -- UNPINNED}, ``torchvision.ops.nms`` = greedy IoU suppression (UNPINNED)."""
    import types
    import facelib_oracle as FO
    deps = os.path.join(REF, 'modules', 'deps')
    for name, sub in (('wm_facelib', ''), ('wm_facelib.detection', 'detection'), ('wm_facelib.detection.yolov5face', 'detection/yolov5face'),
                      ('wm_facelib.detection.yolov5face.models', 'detection/yolov5face/models'),
                      ('wm_facelib.detection.yolov5face.utils', 'detection/yolov5face/utils')):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(deps, 'wm_facelib', sub)]
            sys.modules[name] = m
    cv2 = sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT, cv2.COLOR_BGR2RGB = 1, 0, 4
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    cv2.resize = lambda img, dsize, interpolation=1: FO.cv2_resize_linear_u8(img, int(dsize[0]), int(dsize[1]))
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=(0, 0, 0): np.stack(
        [np.pad(img[:, :, c], ((t, b), (l, r)), constant_values=value[c]) for c in range(3)], 2)
    tv = sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
    tv.ops = types.SimpleNamespace(nms=lambda boxes, scores, thr: torch.tensor(FO._greedy_nms(boxes.numpy(), scores.numpy(), np.float32(thr)),
                                                                              dtype=torch.int64))
    import importlib
    ver, torch.__version__ = torch.__version__, torch.__version__.split('+')[0]      # (face_detector.py:18 cannot parse a '+rocm' local version)
    try:
        fd = importlib.import_module('wm_facelib.detection.yolov5face.face_detector')
    finally:
        torch.__version__ = ver
    out = {}
    for tag, (frames, pred) in FO.yolo_prepost_inputs().items():
        for conf, iou, name in ((0.7, 0.5, 'default'), (0.97, 0.5, 'helper')):      # detect_faces' defaults; the helper's 0.97 (face_restoration_helper.py:221)
            for n in range(2):
                det = fd.YoloDetector.__new__(fd.YoloDetector)
                det.target_size, det.min_face, det.device = None, 10, 'cpu'
                seen = {}

                def network(x, n=n, seen=seen, pred=pred):
                    seen['x'] = x.clone()
                    return (torch.from_numpy(pred[n:n + 1].copy()),)
                network.stride = torch.tensor([8.0, 16.0, 32.0])
                det.detector = network
                res = det.detect_faces(frames[n].copy(), conf, iou)
                if name == 'default':
                    out[f'{tag}_x{n}'] = np.round(seen['x'][0].numpy() * 255.0).astype(np.uint8)       # (u8 / 255 round-trips exactly)
                    assert np.array_equal(out[f'{tag}_x{n}'].astype(np.float32) / np.float32(255.0), seen['x'][0].numpy())
                out[f'{tag}_{name}_det{n}'] = np.zeros((0, 15), np.int64) if res is None else np.asarray(res).astype(np.int64)
                print(tag, name, n, 'input', tuple(seen['x'].shape), 'faces', None if res is None else res.shape)
    np.savez_compressed(os.path.join(GOLD, 'yolo_prepost.npz'), **out)
    print('yolo_prepost.npz:', {k: v.shape for k, v in out.items()})


def main():
    ref = _load(os.path.join(REF, 'modules', 'deps', 'wm_facelib', 'parsing', 'parsenet.py'), 'ref_parsenet')
    out = {}
    for size in (128, 512):
        net = ref.ParseNet(in_size=size, out_size=size).eval()
        sd = PN.synth_parsenet_state_dict(seed=0, in_size=size, out_size=size)
        spec = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert spec == {k: tuple(v) for k, v in PN.parsenet_state_dict_spec(in_size=size, out_size=size).items()}, 'spec drift'
        net.load_state_dict(sd, strict=True)
        x = op_input(f'parsenet{size}', (2 if size == 128 else 1, 3, size, size))
        with torch.no_grad():
            mask = net(x)[0]
        if size == 128:
            out['parsenet128_classes'] = mask.argmax(1).numpy().astype(np.uint8)
            out['parsenet128_logit_grid'] = mask[:, :, 1::4, 2::4].numpy().astype(np.float32)
            top2 = mask.topk(2, dim=1).values
            out['parsenet128_margin'] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
        else:
            out['parsenet512_classes'] = mask.argmax(1).numpy().astype(np.uint8)
            out['parsenet512_logit_grid'] = mask[:, :, 3::16, 5::16].numpy().astype(np.float32)
            top2 = mask.topk(2, dim=1).values
            out['parsenet512_margin'] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
        print(f'ParseNet({size}): logits range', float(mask.min()), float(mask.max()))
    out.update(retinaface_golden())
    out.update(yolo_golden())
    np.savez_compressed(os.path.join(GOLD, 'facelib.npz'), **out)
    print('facelib.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    if '--yolo-prepost' in sys.argv:
        yolo_prepost_golden()
    else:
        main()
