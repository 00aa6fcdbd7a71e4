"""TEST INFRASTRUCTURE ONLY.  Generates ``tests/golden/*.npz`` by running the IMPORTED REFERENCE
(wildminder/ComfyUI-KEEP under /root/reference) on deterministic synthetic weights/inputs.
Runs only in the build container (the reference never travels); the fixtures it writes are
data -- inputs are regenerated from ``engine/synth.py`` seeds, expected outputs are stored.

    python oracle/make_golden.py            # ~1-2 min of CPU

Fixtures
  keep_forward_T3.npz      full KEEP.forward, 'KEEP' config, B=1 T=3 512x512: code indices, logit
                           top-2 margins, Kalman gains, per-frame output digests (32x32 strided grid +
                           per-channel mean/std/min/max), flow digest
  keep_forward_asian_T2.npz  same for the 'Asian' config (cft at 32..256), T=2
  ops.npz                  per-op-class I/O at small sizes (M1-M21 of SURVEY.md 8a.3), outputs of the
                           reference modules; inputs come from ``op_input(name, shape)``
  arch_spec.json           names + shapes of the reference state_dict for both configs
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_import import import_reference_keep, import_reference_module  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import arch, synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
ASIAN = {'cft_list': ['32', '64', '128', '256'], 'temp_reg_list': []}


def op_input(name, shape, scale=1.0):
    """Deterministic test input shared by make_golden, the oracle tests and the GPU tests."""
    n = int(np.prod(shape))
    return torch.from_numpy((synth.uniform_pm1(f'op_input:{name}', n, 7) * scale).astype(np.float32).reshape(shape))


def sub_state(W, prefix):
    pl = len(prefix) + 1
    return {k[pl:]: v for k, v in W.items() if k.startswith(prefix + '.')}


def digest(frames):
    """frames [T,C,H,W] -> strided grid [T,C,32,32] + stats [T,C,4]."""
    T, C, H, Wd = frames.shape
    grid = frames[:, :, 7::H // 32, 5::Wd // 32][:, :, :32, :32].contiguous()
    flat = frames.reshape(T, C, -1)
    stats = torch.stack([flat.mean(-1), flat.std(-1), flat.min(-1).values, flat.max(-1).values], dim=-1)
    return grid.numpy(), stats.numpy()


def full_forward(KEEP, cfg_over, T, fname, wide=False):
    """wide: round 3's regime (i.i.d. flownet weights + the plane-wave clip: flows of hundreds of pixels), kept as the
    out-of-range edge case of the warp path; default: the physical regime of engine/synth.py."""
    cfg = dict(arch.DEFAULT_ARCH, **cfg_over)
    W = synth.synth_state_dict(cfg, seed=0, flow_regime='wide' if wide else 'physical')
    net = KEEP(**cfg).eval()
    net.load_state_dict(W, strict=True)
    x = synth.synth_clip(T=T, B=1, seed=1234, pattern='waves' if wide else 'texture')
    cap = {'logits': [], 'gains': None, 'flows': None}
    net.idx_pred_layer.register_forward_hook(lambda m, i, o: cap['logits'].append(o.detach().permute(1, 0, 2).clone()))
    orig_gain = net.kalman_filter.calc_gain
    orig_flow = net.get_flow

    def gain_spy(z):
        g = orig_gain(z)
        cap['gains'] = g.detach().clone()
        return g

    def flow_spy(xx):
        f = orig_flow(xx)
        cap['flows'] = f.detach().clone()
        return f

    net.kalman_filter.calc_gain = gain_spy
    net.get_flow = flow_spy
    with torch.no_grad():
        out = net(x, need_upscale=False)
    logits = torch.stack(cap['logits'], 1)[0]                       # [T,256,1024]
    top2 = logits.topk(2, dim=-1)
    grid, stats = digest(out[0])
    fgrid, fstats = digest(cap['flows'][0])
    np.savez_compressed(
        os.path.join(GOLD, fname),
        T=T, clip_seed=1234, weight_seed=0,
        indices=top2.indices[..., 0].numpy().astype(np.int16),
        margins=(top2.values[..., 0] - top2.values[..., 1]).numpy().astype(np.float32),
        logit_top1=top2.values[..., 0].numpy().astype(np.float32),
        gains=cap['gains'][0, :, 0].reshape(T, -1).numpy().astype(np.float32),
        out_grid=grid.astype(np.float32), out_stats=stats.astype(np.float32),
        flow_grid=fgrid.astype(np.float32), flow_stats=fstats.astype(np.float32))
    mag = cap['flows'][0].pow(2).sum(1).sqrt().flatten()
    q = [float(mag.kthvalue(max(1, int(mag.numel() * f))).values) for f in (0.5, 0.9, 0.99)]
    print(fname, 'out range', float(out.min()), float(out.max()), 'min margin', float((top2.values[..., 0] - top2.values[..., 1]).min()),
          '| |flow| median %.2f p90 %.2f p99 %.2f max %.2f px, above 8 px: %.2f %%' % (*q, float(mag.max()), 100 * float((mag > 8).float().mean())))
    return W


def per_op(KEEP, W):
    VQ = import_reference_module('wm_basicsr.archs.vqgan_arch')
    KA = import_reference_module('wm_basicsr.archs.keep_arch')
    AU = import_reference_module('wm_basicsr.archs.arch_util')
    GMF = import_reference_module('wm_basicsr.archs.gmflow_arch')
    out = {}

    def run(mod, prefix, *inputs, **kw):
        mod.load_state_dict(sub_state(W, prefix), strict=True)
        mod.eval()
        with torch.no_grad():
            return mod(*inputs, **kw)

    # M1 ResBlock, Cin==Cout and Cin!=Cout (with 1x1 shortcut)
    out['res_same'] = run(VQ.ResBlock(128, 128), 'encoder.blocks.5', op_input('res_same', (1, 128, 16, 16)))
    out['res_proj'] = run(VQ.ResBlock(64, 128), 'encoder.blocks.4', op_input('res_proj', (1, 64, 16, 16)))
    # M2 / M3
    out['down'] = run(VQ.Downsample(128), 'encoder.blocks.6', op_input('down', (1, 128, 16, 16)))
    out['up'] = run(VQ.Upsample(128), 'generator.blocks.17', op_input('up', (1, 128, 8, 8)))
    # M4
    out['attn'] = run(VQ.AttnBlock(512), 'encoder.blocks.17', op_input('attn', (1, 512, 8, 8)))
    # M6 TransformerSALayer (sequence-first)
    pos = W['position_emb'][:64].unsqueeze(1)
    out['sa_layer'] = run(KA.TransformerSALayer(embed_dim=512, nhead=8, dim_mlp=1024, dropout=0.0), 'ft_layers.0',
                          op_input('sa_layer', (64, 1, 512)), query_pos=pos)
    # M8 codebook lookup (one-hot matmul in the reference)
    vqm = VQ.VectorQuantizer(1024, 256, 0.25)
    vqm.load_state_dict(sub_state(W, 'quantize'))
    idx = torch.from_numpy(((synth.uniform_pm1('op_input:codes', 64, 7) + 1) * 512).astype(np.int64).clip(0, 1023))
    with torch.no_grad():
        out['codebook'] = vqm.get_codebook_feat(idx.view(1, 64, 1), shape=[1, 8, 8, 256])
        zq = op_input('vq_nn', (1, 256, 8, 8), 0.7)
        out['vq_nn_idx'] = vqm(zq)[2]['min_encoding_indices'].view(-1).to(torch.int32)
    # M9 CFT
    out['cft'] = run(KA.Fuse_sft_block(256, 256), 'cft.32', op_input('cft_enc', (1, 256, 8, 8)),
                     op_input('cft_dec', (1, 256, 8, 8)), 1)
    # M10 CFA @32 config (C=256, 4 heads x 256) on a 16x16 map
    out['cfa'] = run(KA.CrossFrameFusionLayer(dim=256, num_attention_heads=4, attention_head_dim=256), 'cfa.32',
                     op_input('cfa_curr', (1, 256, 8, 8)), op_input('cfa_prev', (1, 256, 8, 8)))
    # M12 + M13 Kalman gain over a T=3 clip
    kf = KA.KalmanFilter(emb_dim=256, num_attention_heads=8, attention_head_dim=48, num_uncertainty_layers=3)
    kf.load_state_dict(sub_state(W, 'kalman_filter'), strict=True)
    kf.eval()
    with torch.no_grad():
        out['kalman_gain'] = kf.calc_gain(op_input('kalman_z', (1, 3, 256, 8, 8)))
        out['kalman_update'] = kf.update(op_input('ku_z', (1, 256, 8, 8)), op_input('ku_zp', (1, 256, 8, 8)),
                                         (op_input('ku_g', (1, 1, 8, 8)) + 1) / 2)
    # M14 flow warp incl. out-of-range samples
    img = op_input('warp_img', (2, 3, 32, 32))
    flo = op_input('warp_flow', (2, 32, 32, 2), 6.0)
    out['warp'] = AU.flow_warp(img, flo)
    # M16-M21 GMFlow on 64x64 images (8x8 features, 4x4 windows, shift 2)
    fg = GMF.FlowGenerator()
    fg.load_state_dict(sub_state(W, 'flownet'), strict=True)
    fg.eval()
    a = synth.synth_clip(T=2, B=1, size=64, seed=99)[0]
    with torch.no_grad():
        out['gmflow64'] = fg(a[1:2], a[0:1])
    # encoder stack on a 64x64 crop-sized input (all 25 blocks; final map 2x2)
    enc = VQ.Encoder(3, 64, 256, [1, 2, 2, 4, 4, 8], 2, 512, [16])
    # resolution arg only decides where AttnBlocks go; at 64x64 input the same blocks run on smaller maps
    out['encoder64'] = run(enc, 'encoder', op_input('encoder64', (1, 3, 64, 64)))
    gen = VQ.Generator(64, 256, [1, 2, 2, 4, 4, 8], 2, 512, [16])
    out['generator_2x2'] = run(gen, 'generator', op_input('generator_2x2', (1, 256, 2, 2), 0.7))
    # P4 converters: the reference's own img2tensor / tensor2img (img_util.py:9-94) as keep_processor.py:258-259,272 calls them
    IU = import_reference_module('wm_basicsr.utils.img_util')
    crops = [synth.ramp_image(64, 64), np.ascontiguousarray(synth.ramp_image(64, 64)[::-1])]
    ts = [IU.img2tensor(c / 255., bgr2rgb=True, float32=True) for c in crops]
    out['conv_in_crops'] = torch.stack([(t - 0.5) / 0.5 for t in ts])      # torchvision normalize(t, (0.5,)*3, (0.5,)*3)
    xo = op_input('t2i', (2, 3, 64, 64), 1.3)
    xo[0, :, 0, :8] = torch.tensor([-1.2, -1.0, -0.5 / 255, 0.0, 1.0 / 255, 1.0, 1.3, 0.00392156862])
    out['conv_out_u8'] = torch.from_numpy(np.stack([IU.tensor2img(xo[n].clone(), rgb2bgr=True, min_max=(-1, 1)) for n in range(2)]))
    np.savez_compressed(os.path.join(GOLD, 'ops.npz'), **{k: v.numpy() for k, v in out.items()})
    print('ops.npz:', {k: tuple(v.shape) for k, v in out.items()})


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    KEEP = import_reference_keep()
    spec = {}
    for name, over in (('KEEP', {}), ('Asian', ASIAN)):
        net = KEEP(**dict(arch.DEFAULT_ARCH, **over))
        spec[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(GOLD, 'arch_spec.json'), 'w') as f:
        json.dump(spec, f)
    if '--ops-only' in sys.argv:             # per-op goldens only (seconds)
        per_op(KEEP, synth.synth_state_dict(dict(arch.DEFAULT_ARCH), seed=0))
        return
    W = full_forward(KEEP, {}, 3, 'keep_forward_T3.npz')
    per_op(KEEP, W)
    full_forward(KEEP, ASIAN, 2, 'keep_forward_asian_T2.npz')
    # the metric's own clip length: 19 recurrent steps of prev_out -> warp -> hq_encoder -> indices (KA:1062-1127)
    full_forward(KEEP, {}, 20, 'keep_forward_T20.npz')
    full_forward(KEEP, {}, 3, 'keep_forward_T3_wide.npz', wide=True)


if __name__ == '__main__':
    main()
