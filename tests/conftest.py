import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

from __graft_entry__ import load_package  # noqa: E402

load_package()

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _reset_default_ops_overrides():
    """Kernel-vs-kernel tests set ``ops.DEFAULT.flags`` (keep_conv2d_args.flags overrides); a failing test must not leak them."""
    yield
    from comfyui_keep_amd.engine import ops
    ops.DEFAULT.flags, ops.DEFAULT.attn_flags = ops.DEFAULT_CONV_FLAGS, 0


@pytest.fixture(scope='session')
def synth_weights():
    from comfyui_keep_amd.engine import synth
    return synth.synth_state_dict(seed=0)


@pytest.fixture(scope='session', params=['x3', 'fp32'])
def gpu_net(request, synth_weights):
    """The HIP engine with synthetic weights resident on cuda:0 (fails loudly without the library / a gfx950), once per
    parity-grade precision policy: 'x3' (split fp16 operands, the default) and 'fp32' (exact f32 MFMA) must pass the
    SAME assertions."""
    import torch
    from comfyui_keep_amd.engine.net import KeepNet
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    net = KeepNet(**DEFAULT_ARCH)
    net.load_state_dict(synth_weights, strict=True)
    net.to('cuda').eval().set_precision(request.param)
    net._activate_precision()        # module-level tests call the building blocks directly
    return net


def op_input(name, shape, scale=1.0):
    """Same generator as oracle/make_golden.py:op_input (inputs are regenerated, never stored)."""
    import numpy as np
    import torch
    from comfyui_keep_amd.engine import synth
    n = int(np.prod(shape))
    return torch.from_numpy((synth.uniform_pm1(f'op_input:{name}', n, 7) * scale).astype(np.float32).reshape(shape))
