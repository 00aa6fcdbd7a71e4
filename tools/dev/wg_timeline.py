"""dev: phase timeline of the Winograd microkernel (libwinograd_probe_tl.so, -DWG_TL=1): cycles of wave 0 per phase, averaged over blocks."""
import os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import subprocess
import winograd_probe as WP
if not os.path.exists(os.path.join(HERE, 'libwinograd_probe_tl.so')):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-DWG_TL=1', os.path.join(HERE, 'winograd_probe.hip'), '-o',
                           os.path.join(HERE, 'libwinograd_probe_tl.so')])
lib = WP.load('_tl')
torch.manual_seed(0)
NAMES = ['prologue (first tile, chunk 0)', 'first halves (24 MFMAs + activation)', 'barrier 1', 'second halves: groups 5-9 (pass 1)', 'barrier 2',
         'epilogues (two rounds through one stage)', 'second halves: fetch + window loads + group 0 (rows, pass 0)', 'second halves: groups 1-4 (pass 0 positions)']
for (N, hw, cin, cout) in ((8, 512, 64, 64), (8, 256, 128, 128)):
    w = (torch.randn(cout, 3, 3, cin) * 0.05).cuda(); bias = torch.zeros(cout).cuda()
    scale = (torch.rand(N, cin) + 0.5).cuda(); shift = (torch.randn(N, cin) * 0.2).cuda()
    x = torch.randn(N, hw, hw, cin, device='cuda')
    u, inv = WP.pack_weights(w)
    dbg = torch.zeros(16, dtype=torch.int64, device='cuda')
    for _ in range(2):
        dbg.zero_()
        WP.run(lib, x, u, bias, scale, shift, inv, dbg=dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().tolist(); nb = d[8]; tot = sum(d[:8])
    print(f'{N} x {hw}^2 x {cin} -> {cout}: {nb} blocks, {tot / nb:.0f} s_memtime ticks per block (s_memtime ticks of wave 0, ~2.05 per ns here)')
    for q in range(8):
        print(f'   {NAMES[q]:60s} {d[q] / nb:9.0f}  {100.0 * d[q] / tot:5.1f} %')
