"""Static description of the KEEP network topology (hyper-parameters -> block lists
-> state-dict tensor names/shapes).

This is the build's own statement of what the reference constructs in
``KEEP.__init__`` (reference keep_arch.py:862-973), ``Encoder.__init__``
(vqgan_arch.py:247-286), ``Generator.__init__`` (vqgan_arch.py:296-337) and
``GMFlow.__init__`` (gmflow/gmflow.py:13-49).  The released checkpoint is loaded with
``strict=True`` (keep_model_loader.py:120), so the 896 names/shapes produced by
``state_dict_spec`` are the contract; tests/test_arch_spec.py pins them against a
committed listing generated from the imported reference.
"""
from collections import OrderedDict

DEFAULT_ARCH = {
    'img_size': 512, 'emb_dim': 256, 'dim_embd': 512, 'n_head': 8, 'n_layers': 9,
    'codebook_size': 1024, 'cft_list': ['16', '32', '64'], 'kalman_attn_head_dim': 48,
    'num_uncertainty_layers': 3, 'cfa_list': ['16', '32'], 'cfa_nhead': 4, 'cfa_dim': 256, 'cond': 1,
    'nf': 64, 'ch_mult': [1, 2, 2, 4, 4, 8], 'attn_resolutions': [16], 'res_blocks': 2,
    'quantizer_type': 'nearest', 'beta': 0.25, 'temp_reg_list': ['32'], 'latent_size': 256,
}

# feature-size -> channel table and tap positions (keep_arch.py:940-954)
CHANNELS = {'16': 512, '32': 256, '64': 256, '128': 128, '256': 128, '512': 64}
FUSE_ENCODER_BLOCK = {'512': 2, '256': 5, '128': 8, '64': 11, '32': 14, '16': 18}
FUSE_GENERATOR_BLOCK = {'16': 6, '32': 9, '64': 12, '128': 15, '256': 18, '512': 21}

GMFLOW = dict(feature_channels=128, num_layers=6, ffn_dim_expansion=4, upsample_factor=8,
              backbone_dims=(64, 96, 128))


def encoder_blocks(cfg):
    """[(kind, cin, cout)] for Encoder.blocks; kinds: conv, res, attn, down, norm."""
    nf, ch_mult, nres = cfg['nf'], cfg['ch_mult'], cfg['res_blocks']
    res = cfg['img_size']
    in_mult = (1,) + tuple(ch_mult)
    blocks = [('conv', 3, nf)]
    cin = nf
    for i in range(len(ch_mult)):
        cin = nf * in_mult[i]
        cout = nf * ch_mult[i]
        for _ in range(nres):
            blocks.append(('res', cin, cout))
            cin = cout
            if res in cfg['attn_resolutions']:
                blocks.append(('attn', cin, cin))
        if i != len(ch_mult) - 1:
            blocks.append(('down', cin, cin))
            res //= 2
    blocks += [('res', cin, cin), ('attn', cin, cin), ('res', cin, cin),
               ('norm', cin, cin), ('conv', cin, cfg['emb_dim'])]
    return blocks


def generator_blocks(cfg):
    """[(kind, cin, cout)] for Generator.blocks; kinds: conv, res, attn, up, norm."""
    nf, ch_mult, nres = cfg['nf'], cfg['ch_mult'], cfg['res_blocks']
    nres_lv = len(ch_mult)
    cin = nf * ch_mult[-1]
    res = cfg['img_size'] // 2 ** (nres_lv - 1)
    blocks = [('conv', cfg['emb_dim'], cin), ('res', cin, cin), ('attn', cin, cin), ('res', cin, cin)]
    for i in reversed(range(nres_lv)):
        cout = nf * ch_mult[i]
        for _ in range(nres):
            blocks.append(('res', cin, cout))
            cin = cout
            if res in cfg['attn_resolutions']:
                blocks.append(('attn', cin, cin))
        if i != 0:
            blocks.append(('up', cin, cin))
            res *= 2
    blocks += [('norm', cin, cin), ('conv', cin, 3)]
    return blocks


def _resblock_spec(p, cin, cout, out):
    out[f'{p}.norm1.weight'] = (cin,)
    out[f'{p}.norm1.bias'] = (cin,)
    out[f'{p}.conv1.weight'] = (cout, cin, 3, 3)
    out[f'{p}.conv1.bias'] = (cout,)
    out[f'{p}.norm2.weight'] = (cout,)
    out[f'{p}.norm2.bias'] = (cout,)
    out[f'{p}.conv2.weight'] = (cout, cout, 3, 3)
    out[f'{p}.conv2.bias'] = (cout,)
    if cin != cout:
        out[f'{p}.conv_out.weight'] = (cout, cin, 1, 1)
        out[f'{p}.conv_out.bias'] = (cout,)


def _vq_stack_spec(prefix, blocks, out):
    for i, (kind, cin, cout) in enumerate(blocks):
        p = f'{prefix}.blocks.{i}'
        if kind == 'conv':
            out[f'{p}.weight'] = (cout, cin, 3, 3)
            out[f'{p}.bias'] = (cout,)
        elif kind == 'res':
            _resblock_spec(p, cin, cout, out)
        elif kind == 'attn':
            out[f'{p}.norm.weight'] = (cin,)
            out[f'{p}.norm.bias'] = (cin,)
            for n in ('q', 'k', 'v', 'proj_out'):
                out[f'{p}.{n}.weight'] = (cin, cin, 1, 1)
                out[f'{p}.{n}.bias'] = (cin,)
        elif kind in ('down', 'up'):
            out[f'{p}.conv.weight'] = (cin, cin, 3, 3)
            out[f'{p}.conv.bias'] = (cin,)
        elif kind == 'norm':
            out[f'{p}.weight'] = (cin,)
            out[f'{p}.bias'] = (cin,)
        else:
            raise ValueError(kind)


def _ln(p, d, out):
    out[f'{p}.weight'] = (d,)
    out[f'{p}.bias'] = (d,)


def _cross_attn_spec(p, qdim, inner, out):
    for n in ('to_q', 'to_k', 'to_v'):
        out[f'{p}.{n}.weight'] = (inner, qdim)
    out[f'{p}.to_out.0.weight'] = (qdim, inner)
    out[f'{p}.to_out.0.bias'] = (qdim,)


def _geglu_ff_spec(p, d, out):
    out[f'{p}.net.0.proj.weight'] = (8 * d, d)
    out[f'{p}.net.0.proj.bias'] = (8 * d,)
    out[f'{p}.net.2.weight'] = (d, 4 * d)
    out[f'{p}.net.2.bias'] = (d,)


def _gmflow_spec(prefix, out):
    c = GMFLOW['feature_channels']
    d0, d1, d2 = GMFLOW['backbone_dims']
    b = f'{prefix}.backbone'
    out[f'{b}.conv1.weight'] = (d0, 3, 7, 7)
    cin = d0
    for li, (dim, stride) in enumerate(((d0, 1), (d1, 2), (d2, 2)), start=1):
        for bi in range(2):
            p = f'{b}.layer{li}.{bi}'
            bin_ = cin if bi == 0 else dim
            out[f'{p}.conv1.weight'] = (dim, bin_, 3, 3)
            out[f'{p}.conv2.weight'] = (dim, dim, 3, 3)
            if bi == 0 and (stride != 1 or bin_ != dim):
                out[f'{p}.downsample.0.weight'] = (dim, bin_, 1, 1)
                out[f'{p}.downsample.0.bias'] = (dim,)
        cin = dim
    out[f'{b}.conv2.weight'] = (c, d2, 1, 1)
    out[f'{b}.conv2.bias'] = (c,)
    t = f'{prefix}.transformer.layers'
    for i in range(GMFLOW['num_layers']):
        for sub in ('self_attn', 'cross_attn_ffn'):
            p = f'{t}.{i}.{sub}'
            for n in ('q_proj', 'k_proj', 'v_proj', 'merge'):
                out[f'{p}.{n}.weight'] = (c, c)
            _ln(f'{p}.norm1', c, out)
            if sub == 'cross_attn_ffn':
                e = GMFLOW['ffn_dim_expansion']
                out[f'{p}.mlp.0.weight'] = (2 * c * e, 2 * c)
                out[f'{p}.mlp.2.weight'] = (c, 2 * c * e)
                _ln(f'{p}.norm2', c, out)
    f = f'{prefix}.feature_flow_attn'
    for n in ('q_proj', 'k_proj'):
        out[f'{f}.{n}.weight'] = (c, c)
        out[f'{f}.{n}.bias'] = (c,)
    u = f'{prefix}.upsampler'
    out[f'{u}.0.weight'] = (256, 2 + c, 3, 3)
    out[f'{u}.0.bias'] = (256,)
    k = GMFLOW['upsample_factor'] ** 2 * 9
    out[f'{u}.2.weight'] = (k, 256, 1, 1)
    out[f'{u}.2.bias'] = (k,)


def state_dict_spec(cfg=None):
    """OrderedDict name -> shape of every tensor ``KEEP(**cfg).state_dict()`` holds."""
    cfg = dict(DEFAULT_ARCH, **(cfg or {}))
    out = OrderedDict()
    E, D = cfg['emb_dim'], cfg['dim_embd']
    out['position_emb'] = (cfg['latent_size'], D)
    _gmflow_spec('flownet.model', out)
    # Kalman filter (keep_arch.py:751-772)
    inner = cfg['n_head'] * cfg['kalman_attn_head_dim']
    for i in range(cfg['num_uncertainty_layers']):
        p = f'kalman_filter.uncertainty_estimator.{i}'
        _cross_attn_spec(f'{p}.attn1', E, inner, out)
        _ln(f'{p}.norm1', E, out)
        _geglu_ff_spec(f'{p}.ff', E, out)
        _ln(f'{p}.norm3', E, out)
        _cross_attn_spec(f'{p}.attn_temp', E, inner, out)
        _ln(f'{p}.norm_temp', E, out)
    for i in range(3):
        _resblock_spec(f'kalman_filter.kalman_gain_calculator.{i}', E, E, out)
    out['kalman_filter.kalman_gain_calculator.3.weight'] = (1, E, 1, 1)
    out['kalman_filter.kalman_gain_calculator.3.bias'] = (1,)
    _vq_stack_spec('hq_encoder', encoder_blocks(cfg), out)
    _vq_stack_spec('encoder', encoder_blocks(cfg), out)
    out['quantize.embedding.weight'] = (cfg['codebook_size'], E)
    _vq_stack_spec('generator', generator_blocks(cfg), out)
    out['feat_emb.weight'] = (D, E)
    out['feat_emb.bias'] = (D,)
    for i in range(cfg['n_layers']):
        p = f'ft_layers.{i}'
        out[f'{p}.self_attn.in_proj_weight'] = (3 * D, D)
        out[f'{p}.self_attn.in_proj_bias'] = (3 * D,)
        out[f'{p}.self_attn.out_proj.weight'] = (D, D)
        out[f'{p}.self_attn.out_proj.bias'] = (D,)
        out[f'{p}.linear1.weight'] = (2 * D, D)
        out[f'{p}.linear1.bias'] = (2 * D,)
        out[f'{p}.linear2.weight'] = (D, 2 * D)
        out[f'{p}.linear2.bias'] = (D,)
        _ln(f'{p}.norm1', D, out)
        _ln(f'{p}.norm2', D, out)
    _ln('idx_pred_layer.0', D, out)
    out['idx_pred_layer.1.weight'] = (cfg['codebook_size'], D)
    for sz in cfg['cfa_list']:
        C = CHANNELS[sz]
        p = f'cfa.{sz}'
        _ln(f'{p}.norm1', C, out)
        _ln(f'{p}.norm2', C, out)
        _geglu_ff_spec(f'{p}.ff', C, out)
        _cross_attn_spec(f'{p}.attn', C, cfg['cfa_nhead'] * cfg['cfa_dim'], out)
    for sz in cfg['cft_list']:
        C = CHANNELS[sz]
        p = f'cft.{sz}'
        _resblock_spec(f'{p}.encode_enc', 2 * C, C, out)
        for br in ('scale', 'shift'):
            for j in (0, 2):
                out[f'{p}.{br}.{j}.weight'] = (C, C, 3, 3)
                out[f'{p}.{br}.{j}.bias'] = (C,)
    return out
