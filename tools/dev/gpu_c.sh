mkdir -p gpurun_out/r2j
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^  File\|^Extension" | tail -15 > gpurun_out/r2j/tests.log
tail -12 gpurun_out/r2j/tests.log
bash tools/profile_step.sh x3 16 r2j > gpurun_out/r2j/profile_x3_16.out 2>&1; tail -32 gpurun_out/r2j/profile_x3_16.out
cd /tmp && export TMPDIR=/tmp && KEEP_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2j/trace_x3_b1 -o t -- python /root/repo/tools/run_step.py x3 1 3 > /root/repo/gpurun_out/r2j/trace_b1.log 2>&1
python profiles/summarize_rocpd.py $(find gpurun_out/r2j/trace_x3_b1 -name "*results.db" | head -1) 3 > gpurun_out/r2j/x3_b1_kernel_stats.txt; head -24 gpurun_out/r2j/x3_b1_kernel_stats.txt
for g in 0 1; do KEEP_AMD_GRAPH=$g python - <<'PY'
import sys, time, torch, os
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
net = KeepNet(**DEFAULT_ARCH); net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval()
for prec in ('x3', 'bf16'):
    net.set_precision(prec)
    for B in (1, 2):
        x = synth.synth_clip(T=20, B=B, seed=1234).cuda()
        net(x); net(x); torch.cuda.synchronize()
        t0 = time.perf_counter(); net(x); net(x); net(x); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print('graph', os.environ['KEEP_AMD_GRAPH'], prec, 'B', B, 'ms/clip-batch', round(dt * 1e3, 1), 'frames/s', round(B * 20 / dt, 1))
PY
done
