"""dev / evidence: the Winograd F(2x2, 3x3) x3 microkernel of tools/dev/winograd_probe.hip against the product's streaming kernel
on the dominant shapes -- numerics against an fp64 convolution of the same activated input, time per launch by HIP events over a hipGraph of launches.

    python tools/dev/winograd_probe.py [N]          # N images per launch (default 48, the bench's batch)

Table for profiles/r06_winograd_microkernel.txt.  The microkernel has the dominant instantiation's prologue (GroupNorm affine + swish) and a bias
epilogue; it has NO residual, NO GroupNorm statistics, NO max|out| (the product kernel is timed with and without those)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)


def load(variant=''):
    lib = ctypes.CDLL(os.path.join(HERE, f'libwinograd_probe{variant}.so'))
    lib.wg_probe_run.restype = ctypes.c_int
    lib.wg_probe_run.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return lib


def pack_weights(w):
    """w [Cout, 3, 3, Cin] fp32 (the product's layout) -> (fragment-ordered hi/lo fp16 words [cb][chunk][position 16][cout half 2][hi, lo][lane 64][8], 1 / scale):
    U = G g G^T in fp64, rounded to fp32, scaled by a power of two just below the fp16 range, split hi = f16(U s), lo = f16(U s - hi)."""
    Cout, _, _, Cin = w.shape
    U = torch.einsum('ij,ojkc,lk->oilc', G, w.double().cpu(), G).float()                  # [Cout, i, l, Cin]
    sc = 2.0 ** (14 - int(torch.ceil(torch.log2(U.abs().max())).item()))
    Us = U * sc
    hi = Us.half()
    lo = (Us - hi.float()).half()
    ncb, nch = Cout // 64, Cin // 16
    out = torch.empty(ncb, nch, 16, 2, 2, 64, 8, dtype=torch.float16)
    for hl, t in enumerate((hi, lo)):
        # t[co, i, l, ci] -> [cb, ch, l31, p = i * 4 + l, chunk, lhi, e]
        v = t.reshape(ncb, 2, 32, 16, nch, 2, 8)
        out[:, :, :, :, hl] = v.permute(0, 4, 3, 1, 5, 2, 6).reshape(ncb, nch, 16, 2, 64, 8)      # lane = lhi * 32 + l31
    return out.contiguous().cuda(), 1.0 / sc


def run(lib, x, u, bias, scale, shift, acc_scale, swish=True, dbg=None):
    N, H, W, Cin = x.shape
    Cout = bias.shape[0]
    y = torch.empty(N, H, W, Cout, device='cuda')
    rc = lib.wg_probe_run(x.data_ptr(), u.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), N, H, W, Cin, Cout,
                          acc_scale, 1 if swish else 0, torch.cuda.current_stream().cuda_stream, None if dbg is None else dbg.data_ptr(), 0)
    assert rc == 0, rc
    return y


def reference(x, w, bias, scale, shift):
    """fp64: swish(x * scale + shift) convolved 3x3, zero padding of the activated tensor."""
    a = x.double() * scale.double()[:, None, None, :] + shift.double()[:, None, None, :]
    a = a * torch.sigmoid(a)
    y = torch.nn.functional.conv2d(a.permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1)
    return y.permute(0, 2, 3, 1)


def timed(fn, iters=6):
    """us per launch: best of three replays of a hipGraph of ``iters`` launches, after 12 warm launches (the first kernel timed in a process otherwise pays the clock ramp)."""
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    from __graft_entry__ import load_package
    load_package()
    from comfyui_keep_amd.engine import hiplib as L, ops
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    import glob
    variants = sorted(os.path.basename(f)[len('libwinograd_probe'):-3] for f in glob.glob(os.path.join(HERE, 'libwinograd_probe*.so')) if not f.endswith('_tl.so'))
    libs = {v: load(v) for v in variants}
    torch.manual_seed(0)
    print(f'# python tools/dev/winograd_probe.py {N}   (1 x MI355X; us per launch: best of three replays of a hipGraph of 6 launches after 12 warm launches; TF/s = ALGORITHMIC 2 N H W Cin Cout 9 / time -- the direct '
          f'convolution\'s FLOPs, so the Winograd rows are comparable with the product rows and with roofline.frac)')
    for hw, cin, cout in ((512, 64, 64), (256, 128, 128)):
        w = (torch.randn(cout, 3, 3, cin) * (0.5 / (3 * cin ** 0.5))).cuda()
        bias = (torch.randn(cout) * 0.1).cuda()
        scale = (torch.rand(N, cin) + 0.5).cuda()
        shift = (torch.randn(N, cin) * 0.2).cuda()
        x = torch.randn(N, hw, hw, cin, device='cuda')
        u, inv = pack_weights(w)
        flops = 2.0 * N * hw * hw * cin * cout * 9
        # numerics on the first two images against fp64
        ref = reference(x[:2], w, bias, scale[:2], shift[:2])
        rows = []
        for v, lib in libs.items():
            y = run(lib, x, u, bias, scale, shift, inv)
            err = (y[:2].double() - ref).abs().max().item()
            t = timed(lambda: run(lib, x, u, bias, scale, shift, inv))
            rows.append(('winograd F(2x2,3x3) x3 microkernel' + (' [build ' + v[1:] + ']' if v else ''), t, err))
        # the product's kernel on the same operands: plain (prologue + bias) and as the step runs it (+ residual + GroupNorm statistics + max|out|)
        sc = ops.x3_scale_for(float(w.abs().max()))
        wx3 = ops.split_x3(w.reshape(-1, cin), sc).view(-1)
        res = torch.randn(N, hw, hw, cout, device='cuda')
        kw = dict(mma=L.MMA_X3, wx3=wx3, x3_acc_scale=1.0 / sc, pro=(scale, shift), pro_act=L.PRO_SWISH)
        yp = ops.conv(x, w, bias, **kw)
        errp = (yp[:2].double() - ref).abs().max().item()
        rows.append(('product conv3x3_halo_x3s_kernel, prologue + bias only', timed(lambda: ops.conv(x, w, bias, **kw)), errp))
        rows.append(('product conv3x3_halo_x3s_kernel, + residual + GroupNorm statistics + max|out| (as in the step)',
                     timed(lambda: ops.conv(x, w, bias, residual=res, stats=True, **kw)), float('nan')))
        print(f'{N} x {hw}^2 x {cin} -> {cout}   (output scale {ref.abs().max().item():.2f})')
        for name, t, err in rows:
            print(f'  {name:100s} {t:9.1f} us  {flops / t / 1e6:7.1f} TF/s algorithmic   max |y - fp64| {err:.3e}')


if __name__ == '__main__':
    main()
