mkdir -p gpurun_out/r2k
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -rf -k "u8_path or hipgraph" 2>&1 | grep -v "^  File\|^Extension" | tail -40
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp && KEEP_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r2k/trace_x3_b1 -o t -- python $REPO/tools/run_step.py x3 1 3 > $REPO/gpurun_out/r2k/trace_b1.log 2>&1
cd $REPO
python profiles/summarize_rocpd.py $(find gpurun_out/r2k/trace_x3_b1 -name "*results.db" | head -1) 3 > gpurun_out/r2k/x3_b1_kernel_stats.txt; head -30 gpurun_out/r2k/x3_b1_kernel_stats.txt
for g in 0 1; do KEEP_AMD_GRAPH=$g python - <<'PY'
import sys, time, torch, os
sys.path.insert(0, os.getcwd())
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
