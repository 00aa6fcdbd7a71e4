"""Phase ablations of the x3 halo kernel with the shader clock / power sampled while each variant runs back to back.
   Needs the dev library:  hipcc ... -DKEEP_X3_ABLATE (tools/dev/README.md); ABL_LIB=<path of that .so>
   python tools/dev/ablate_probe.py [layer ...]      layers: c64_512 c128_256 c256_64"""
import glob
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from comfyui_keep_amd.engine import hiplib as L  # noqa: E402

if os.environ.get('ABL_LIB'):
    L.LIB_PATH = os.environ['ABL_LIB']
from comfyui_keep_amd.engine import ops  # noqa: E402

LAYERS = {'c64_512': (16, 512, 512, 64, 64), 'c128_256': (16, 256, 256, 128, 128), 'c256_64': (16, 64, 64, 256, 256)}
EXPS = [('product', None), ('17 product + counter', 17), ('1 no fragment reads', 1), ('2 no MFMAs', 2), ('3 no staging', 3), ('4 no stores', 4), ('5 no fetch', 5),
        ('10 no swish', 10), ('11 no epilogue', 11), ('12 DMA not waited', 12), ('13 no weight DMA', 13), ('14 no halo ds_write', 14), ('15 DMA from one KB', 15), ('16 no DMA, B from halo', 16),
        ('9 timeline', 9)]
if os.environ.get('ABL_EXPS'):
    want = set(os.environ['ABL_EXPS'].split(','))
    EXPS = [e for e in EXPS if str(e[1] or 0) in want]


def sysfs_nodes():
    out = {}
    for pat, key in (('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input', 'sclk_hz'),
                     ('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average', 'power_uw'),
                     ('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input', 'power_in_uw')):
        g = sorted(glob.glob(pat))
        if g:
            out[key] = g[0]
    return out


NODES = sysfs_nodes()


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            row = {}
            for k, path in NODES.items():
                try:
                    row[k] = float(open(path).read().strip())
                except Exception:
                    pass
            self.rows.append(row)
            time.sleep(0.02)

    def mean(self, key):
        v = [r[key] for r in self.rows[len(self.rows) // 4:] if key in r]      # skip the ramp
        return sum(v) / len(v) if v else float('nan')


def run(name, seconds=1.5):
    N, H, W, Cin, Cout = LAYERS[name]
    x = torch.randn(N, H, W, Cin, device='cuda')
    w = torch.randn(Cout, 3, 3, Cin, device='cuda') * 0.05
    b = torch.randn(Cout, device='cuda')
    sc = ops.x3_scale_for(float(w.abs().max()))
    kw = dict(pad=1, ksize=3, mma=L.MMA_X3, stats=True, wx3=ops.split_x3(w.reshape(-1, Cin), sc).view(-1), x3_acc_scale=1.0 / sc,
              pro=(torch.ones(N, Cin, device='cuda'), torch.zeros(N, Cin, device='cuda')), pro_act=L.PRO_SWISH)
    flops = 2.0 * N * H * W * Cin * Cout * 9
    for label, e in EXPS:
        for per_cu in ((2, 1) if e is None else (2,)):
            os.environ.pop('KEEP_X3_EXP', None)
            if e is not None:
                os.environ['KEEP_X3_EXP'] = str(e)
            os.environ['KEEP_X3_BLOCKS_PER_CU'] = str(per_cu)
            for _ in range(3):
                ops.conv(x, w, b, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv(x, w, b, **kw)
            e1.record()
            torch.cuda.synchronize()
            one = e0.elapsed_time(e1)
            iters = max(10, int(seconds * 1e3 / max(one, 1e-3)))
            smp = Sampler()
            smp.start()
            e0.record()
            for _ in range(iters):
                ops.conv(x, w, b, **kw)
            e1.record()
            torch.cuda.synchronize()
            smp.stop = True
            smp.join()
            ms = e0.elapsed_time(e1) / iters
            if e is not None:            # the LAST launch of a back-to-back run with the per-block cycle / 100 MHz counters read back
                for _ in range(iters // 2):
                    ops.conv(x, w, b, **kw)
                os.environ['KEEP_X3_CYC'] = '1'
                ops.conv(x, w, b, **kw)
                torch.cuda.synchronize()
                os.environ.pop('KEEP_X3_CYC')
            print(f'{name:9s} {label:20s} blocks/CU {per_cu}  {ms * 1e3:8.1f} us  {flops / ms / 1e9:6.1f} TF(nominal)  '
                  f'sclk {smp.mean("sclk_hz") / 1e6:7.0f} MHz  power {smp.mean("power_in_uw") / 1e6:6.0f} W  ({len(smp.rows)} samples)', flush=True)
    os.environ.pop('KEEP_X3_EXP', None)
    os.environ.pop('KEEP_X3_BLOCKS_PER_CU', None)


if __name__ == '__main__':
    print('sysfs nodes:', NODES, flush=True)
    for n in (sys.argv[1:] or ['c64_512', 'c128_256']):
        run(n)
