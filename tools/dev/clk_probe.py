import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import hiplib as L, ops
sys.argv = ['x']
os.environ['X3'] = '1'
import importlib.util
spec = importlib.util.spec_from_file_location('bc', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench_conv.py'))
bc = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(bc)
except SystemExit:
    pass
pro = len(sys.argv) > 99
bc.run('c128_256', L.MMA_X3, os.environ.get('PRO') is not None, iters=1)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * 8192)()
lib.keep_debug_read_clk.restype = ctypes.c_int
print('rc', lib.keep_debug_read_clk(buf, 8192))
t0 = buf[0]
for iv in range(0, 40):
    row = []
    for grp in range(2):
        b = (iv * 2 + grp) * 4
        row.append((buf[b] - t0, buf[b + 1] - buf[b], buf[b + 2] - buf[b + 1]))
    print(iv, 'g0 start %7d work %6d barrier-wait %6d | g1 start %7d work %6d barrier-wait %6d' % (row[0] + row[1]))
