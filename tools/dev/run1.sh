cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -x -q -m gpu -k "fused_range_maxima or cfa_range or layernorm or geglu or cfa" 2>&1 | tail -5
timeout 900 python tools/dev/cfa_ab.py 1 2>&1 | grep "^B="
timeout 900 python tools/dev/cfa_ab.py 16 2>&1 | grep "^B="
