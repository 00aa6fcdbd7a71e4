cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_x3_latency or linear" 2>&1 | tail -3
timeout 600 python tools/dev/gemm_lat_bench.py 2>&1 | grep -E " 16:|  8:| hw K" | head -14
timeout 600 python tools/dev/flag_ab.py 16 new=0 2>&1 | tail -3
