set -x
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2a/kernels.log
X3=1 timeout 300 python tools/bench_conv.py c64_512 c128_256 up128_512 c256_64 c512_16 lin128 lin512_1024 > gpurun_out/r2a/bench_x3.log 2>&1
timeout 300 python tools/bench_conv.py c64_512 c128_256 up128_512 > gpurun_out/r2a/bench_bf16.log 2>&1
F32=1 timeout 300 python tools/bench_conv.py c64_512 c128_256 > gpurun_out/r2a/bench_f32.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_net.py -m gpu -q -s 2>&1 | tail -80 > gpurun_out/r2a/net.log
for p in x3 fp32; do timeout 600 python - $p > gpurun_out/r2a/step_$p.log 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
net = KeepNet(**DEFAULT_ARCH); net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision(sys.argv[1])
for B in (16,):
    x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
    net(x); torch.cuda.synchronize()
    t0 = time.perf_counter(); net(x); net(x); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(sys.argv[1], 'B', B, 'ms/step', dt * 1e3, 'frames/s', B * 20 / dt)
PY
done
cat gpurun_out/r2a/kernels.log gpurun_out/r2a/bench_x3.log gpurun_out/r2a/step_*.log
