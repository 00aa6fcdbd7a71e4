import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L, ops
torch.manual_seed(0)
def mk(cout, cin, k=3, s=0.05):
    w = torch.randn(cout, k, k, cin, device='cuda') * s
    sc = ops.x3_scale_for(float(w.abs().max()))
    return w, ops.split_x3(w.reshape(-1, cin), sc).view(-1), 1.0 / sc
N, H, C = 4, 64, 256
e = torch.randn(N, H, H, C, device='cuda') * 30
w0, w0x, a0 = mk(2 * C, C)
b0 = torch.randn(2 * C, device='cuda')
for mma in (L.MMA_F32, L.MMA_X3):
    kw = dict(mma=mma)
    if mma == L.MMA_X3: kw.update(wx3=w0x, x3_acc_scale=a0)
    ss, st = ops.conv(e, w0, b0, act=L.ACT_LRELU02, stats=True, **kw)
    torch.cuda.synchronize()
    print('mma', mma, 'ss finite', bool(torch.isfinite(ss).all()), 'absmax per image', ss.abs().flatten(1).max(1).values.tolist(),
          'st.amax', None if st is None or st.amax is None else st.amax.tolist())
    if mma == L.MMA_F32: ref = ss
print('x3 vs f32 ss err', float((ss - ref).abs().max()))
w1, w1x, a1 = mk(C, C)
b1 = torch.randn(C, device='cuda')
for off in (0, C):
    y32 = ops.conv(ref, w1, b1, cin=C, in_off=off, mma=L.MMA_F32)
    y3 = ops.conv(ref, w1, b1, cin=C, in_off=off, mma=L.MMA_X3, wx3=w1x, x3_acc_scale=a1)
    y3a = ops.conv(ref, w1, b1, cin=C, in_off=off, mma=L.MMA_X3, wx3=w1x, x3_acc_scale=a1, x_amax=ref.abs().flatten(1).max(1).values.contiguous())
    torch.cuda.synchronize()
    print('off', off, 'finite', bool(torch.isfinite(y3).all()), bool(torch.isfinite(y3a).all()), 'err', float((y3 - y32).abs().max()), float((y3a - y32).abs().max()), 'scale', float(y32.abs().max()))
print('---- isolate')
for name, kw2, x_in in (('simple+probe', dict(), e), ('act+bounded', dict(act=L.ACT_LRELU02, bounded=True), e / 30), ('act+probe', dict(act=L.ACT_LRELU02), e),
                        ('act+probe N=1', dict(act=L.ACT_LRELU02), e[:1].contiguous()), ('relu+probe', dict(act=L.ACT_RELU), e)):
    y32 = ops.conv(x_in, w0, b0, mma=L.MMA_F32, **{k: v for k, v in kw2.items() if k != 'bounded'})
    y3, st = ops.conv(x_in, w0, b0, mma=L.MMA_X3, wx3=w0x, x3_acc_scale=a0, stats=True, **kw2)
    torch.cuda.synchronize()
    d = (y3 - y32).abs()
    bad = d > 1e-2 * float(y32.abs().max())
    print(name, 'err', float(d.max()), 'scale', float(y32.abs().max()), 'bad frac', float(bad.float().mean()),
          'bad per image', bad.flatten(1).float().mean(1).tolist(), 'bad per cout block', bad.reshape(-1, 8, 64).float().mean((0, 2)).tolist())
    if bad.any():
        idx = torch.nonzero(bad)[:4]
        for i in idx.tolist():
            print('   at', i, 'x3', float(y3[tuple(i)]), 'f32', float(y32[tuple(i)]))
