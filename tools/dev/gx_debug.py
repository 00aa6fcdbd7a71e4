"""dev: where does gemm_x3s_kernel differ from conv_x3_kernel?  gx_debug.py M K N [gelu] [bias] [two]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L
if os.environ.get('ABL_LIB'):
    L.LIB_PATH = os.environ['ABL_LIB']
from comfyui_keep_amd.engine import ops
M, K, N = (int(v) for v in sys.argv[1:4])
flags = sys.argv[4:]
torch.manual_seed(0)
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05
b = torch.randn(N, device='cuda') if 'bias' in flags else None
sc = ops.x3_scale_for(float(w.abs().max()))
kw = dict(mma=L.MMA_X3, wx3=ops.split_x3(w, sc).view(-1), x3_acc_scale=1.0 / sc, pad=0, ksize=1, bounded=True,
          act=L.ACT_GELU if 'gelu' in flags else L.ACT_NONE)
def run():
    ops._PLAN_CACHE.clear()
    if 'two' in flags:
        return ops.conv(x[:, :K // 2].contiguous().view(1, M, 1, K // 2), w, b, x2=x[:, K // 2:].contiguous().view(1, M, 1, K // 2), **kw).reshape(M, N)
    return ops.conv(x.view(1, M, 1, K), w, b, **kw).reshape(M, N)
os.environ.pop('KEEP_X3_NO_GEMM_STREAM', None)
ys = run().clone()
os.environ['KEEP_X3_NO_GEMM_STREAM'] = '1'
yt = run().clone()
torch.cuda.synchronize()
bad = (ys != yt)
print(f'M={M} K={K} N={N} {flags}: mismatching elements {int(bad.sum())} of {bad.numel()}  max diff {float((ys - yt).abs().max()):.3e}')
if bad.any():
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print(' bad rows:', len(rows), rows[:24].tolist(), '... tiles (row//128):', sorted(set((rows // 128).tolist()))[:20])
    print(' bad cols:', len(cols), cols[:40].tolist())
    r0 = int(rows[0]); print(' row', r0, 'bad cols', bad[r0].nonzero().flatten()[:40].tolist())
    c0 = int(bad[r0].nonzero()[0]); print(' stream', ys[r0, c0:c0 + 8].tolist(), '\n tile  ', yt[r0, c0:c0 + 8].tolist())
    # is the streamed value some other element of the tile result?
    v = ys[r0, c0]; hit = (yt[r0 // 128 * 128:r0 // 128 * 128 + 128] == v).nonzero()
    print(' value found in the tile result at (row in tile, col):', hit[:8].tolist(), ' (bad element at', r0 % 128, c0, ')')
