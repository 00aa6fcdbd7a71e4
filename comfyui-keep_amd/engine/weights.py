"""Weight ingestion: reference ``state_dict`` (896 fp32 tensors, keep_model_loader.py:99-121) ->
ONE packed fp32 blob in kernel-friendly layouts + an index of named views.

Layouts
  * conv weights ``[Cout,Cin,KH,KW]`` -> ``[Cout,KH,KW,Cin]`` (K-contiguous rows for the implicit GEMM;
    activations are NHWC so both MFMA operands stream along the reduction axis);
  * linear weights stay ``[out,in]``;
  * projections that read the same input are concatenated so they run as one GEMM:
      AttnBlock q|k|v (VQ:190-210)            -> ``<p>.qkv``      [3C, C]
      Kalman attn1 / attn_temp to_q|k|v       -> ``<p>.to_qkv``   [3*inner, C]   (KA:693-702, 145-169)
      CFA to_k|to_v (both read the prev frame)-> ``<p>.to_kv``    [2*inner, C]   (KA:168-169)
      GMFlow q|k|v_proj                       -> ``<p>.qkv``      [3C, C]        (GM/transformer.py:161-163)
      CFT scale.0|shift.0 (both read e)       -> ``<p>.ss0``      [2C,3,3,C]     (KA:468-469)
The blob is what travels: one H2D copy per ``load_device()`` and one RCCL broadcast per node (8e).
"""
import numpy as np
import torch

from .arch import DEFAULT_ARCH, encoder_blocks, generator_blocks, state_dict_spec

_ALIGN = 64  # elements (256 B): every view 16-byte aligned for float4 loads


def validate_state_dict(sd, cfg):
    """strict=True semantics of nn.Module.load_state_dict: exact key set and shapes."""
    spec = state_dict_spec(cfg)
    missing = [k for k in spec if k not in sd]
    unexpected = [k for k in sd if k not in spec]
    bad = [f'{k}: {tuple(sd[k].shape)} vs {tuple(spec[k])}' for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
    if missing or unexpected or bad:
        msg = 'Error(s) in loading state_dict for KEEP:'
        if missing:
            msg += f'\n\tMissing key(s): {missing[:8]}{" ..." if len(missing) > 8 else ""}'
        if unexpected:
            msg += f'\n\tUnexpected key(s): {unexpected[:8]}{" ..." if len(unexpected) > 8 else ""}'
        if bad:
            msg += f'\n\tsize mismatch: {bad[:8]}'
        raise RuntimeError(msg)
    return spec


def _conv_pack(w):
    return w.permute(0, 2, 3, 1).contiguous()


def logical_tensors(sd, cfg=None):
    """name -> fp32 CPU tensor in kernel layout (fused / permuted as described above)."""
    cfg = dict(DEFAULT_ARCH, **(cfg or {}))
    out = {}
    consumed = set()

    def take(name):
        consumed.add(name)
        return sd[name].detach().to(torch.float32).cpu()

    def cat(names):
        return torch.cat([take(n) for n in names], dim=0)

    # --- fused groups
    for prefix, blocks in (('encoder', encoder_blocks(cfg)), ('hq_encoder', encoder_blocks(cfg)),
                           ('generator', generator_blocks(cfg))):
        for i, (kind, cin, _) in enumerate(blocks):
            if kind == 'attn':
                p = f'{prefix}.blocks.{i}'
                out[f'{p}.qkv.weight'] = cat([f'{p}.{n}.weight' for n in 'qkv']).reshape(3 * cin, cin)
                out[f'{p}.qkv.bias'] = cat([f'{p}.{n}.bias' for n in 'qkv'])
    for i in range(cfg['num_uncertainty_layers']):
        for a in ('attn1', 'attn_temp'):
            p = f'kalman_filter.uncertainty_estimator.{i}.{a}'
            out[f'{p}.to_qkv.weight'] = cat([f'{p}.to_{n}.weight' for n in 'qkv'])
    for sz in cfg['cfa_list']:
        p = f'cfa.{sz}.attn'
        out[f'{p}.to_kv.weight'] = cat([f'{p}.to_{n}.weight' for n in 'kv'])
    for sz in cfg['cft_list']:
        p = f'cft.{sz}'
        out[f'{p}.ss0.weight'] = _conv_pack(cat([f'{p}.scale.0.weight', f'{p}.shift.0.weight']))
        out[f'{p}.ss0.bias'] = cat([f'{p}.scale.0.bias', f'{p}.shift.0.bias'])
    for name in list(sd.keys()):
        if name.startswith('flownet.model.transformer.layers.') and name.endswith('.q_proj.weight'):
            p = name[:-len('.q_proj.weight')]
            out[f'{p}.qkv.weight'] = cat([f'{p}.{n}_proj.weight' for n in 'qkv'])
    # --- everything else: conv weights permuted, the rest as is
    for name, t in sd.items():
        if name in consumed:
            continue
        t = t.detach().to(torch.float32).cpu()
        if t.dim() == 4:
            t = _conv_pack(t)
            if t.shape[1] == 1 and t.shape[2] == 1:
                t = t.reshape(t.shape[0], t.shape[3])          # 1x1 conv == linear
            elif t.shape[1] == 3 and t.shape[3] > 16 and t.shape[3] % 16:
                # 3x3 conv with a ragged channel count (GMFlow upsampler.0: cat[flow 2 | feature 128] = 130): zero input
                # columns up to a multiple of 16 -- the net pads the activation the same way (ops.concat2(pad_to=16)) and the
                # layer runs on the 16-channel-chunk MFMA kernels instead of the generic gather kernel
                t = torch.nn.functional.pad(t, (0, 16 - t.shape[3] % 16))
        out[name] = t.contiguous()
    # CFT (KA:465-472): encode_enc is a ResBlock over cat[enc_feat, dec_feat].  Its first convolution and its 1x1 shortcut are linear in
    # their input channels and the GroupNorm in front (32 groups over 2C channels) never mixes the two halves, so the encoder half of both
    # -- which depends on the LQ frame only -- is evaluated once per clip for all frames (engine/net.py:_cft_enc_part) and the frame
    # recurrence runs the decoder half: the [.., :C] / [.., C:] input-channel slices as tensors of their own (derived, like `_s2d`)
    for sz in cfg['cft_list']:
        p = f'cft.{sz}.encode_enc'
        w1, wo = out.get(f'{p}.conv1.weight'), out.get(f'{p}.conv_out.weight')
        if w1 is not None and wo is not None and w1.shape[-1] % 2 == 0 and w1.shape[-1] == wo.shape[-1] and (w1.shape[-1] // 2) % 32 == 0:
            C = w1.shape[-1] // 2
            out[f'{p}.conv1.weight_enc'], out[f'{p}.conv1.weight_dec'] = w1[..., :C].contiguous(), w1[..., C:].contiguous()
            out[f'{p}.conv_out.weight_enc'], out[f'{p}.conv_out.weight_dec'] = wo[..., :C].contiguous(), wo[..., C:].contiguous()
    k7 = 'flownet.model.backbone.conv1.weight'
    if k7 in out and tuple(out[k7].shape[1:]) == (7, 7, 3):
        out[k7 + '_s2d'] = s2d_weights_7x7(out[k7])
    return out


def s2d_weights_7x7(w):
    """Packed [Cout,7,7,3] weights of a stride-2 pad-3 convolution -> [Cout,4,4,16] weights of the same convolution as a stride-1
    convolution (pad 2 top / left) on the 2x2 space-to-depth image of keep_rgb_s2d: input row 2y + ky - 3 = 2(y + Ky - 2) + dy with
    Ky = (ky + 1) // 2, dy = (ky + 1) % 2; channel (dy*2 + dx)*3 + c; (Ky, dy) = (0, 0) and channels 12..15 carry zeros."""
    cout = w.shape[0]
    o = torch.zeros(cout, 4, 4, 16, dtype=w.dtype)
    for ky in range(7):
        for kx in range(7):
            Ky, dy, Kx, dx = (ky + 1) // 2, (ky + 1) % 2, (kx + 1) // 2, (kx + 1) % 2
            o[:, Ky, Kx, (dy * 2 + dx) * 3:(dy * 2 + dx) * 3 + 3] = w[:, ky, kx, :]
    return o


def pack_blob(tensors):
    """dict name->CPU tensor -> (flat fp32 numpy blob, index name -> (offset, shape))."""
    index, off = {}, 0
    for name, t in tensors.items():
        index[name] = (off, tuple(t.shape))
        off += (t.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    blob = np.zeros(off, dtype=np.float32)
    for name, t in tensors.items():
        o, _ = index[name]
        blob[o:o + t.numel()] = t.reshape(-1).numpy()
    return blob, index


def views(blob_t, index):
    """Named views into a (device) blob tensor."""
    out = {}
    for name, (off, shape) in index.items():
        n = int(np.prod(shape))
        out[name] = blob_t[off:off + n].view(shape)
    return out
