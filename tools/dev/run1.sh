cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "comfy" 2>&1 | tail -5
python -m pytest tests/test_gpu_paste.py -x -q -m gpu -k "streamed" 2>&1 | tail -5
python tools/dev/x3_ceiling.py > gpurun_out/r06_x3_ceiling_probe.txt 2>&1; cat gpurun_out/r06_x3_ceiling_probe.txt
python tools/dev/product_leg.py both 2>&1 | grep -v Warning | cut -c1-1500
python -m pytest tests/test_gpu_net.py -x -q -m gpu -s -k "T20_vs_reference or first_flips or T3_vs_reference or asian" 2>&1 | grep -v "^$" | cut -c1-1200 | tail -40
