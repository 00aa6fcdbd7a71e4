set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_x3_latency or linear or downsample or conv_x3_gather" 2>&1 | tail -5
timeout 600 python tools/dev/gemm_lat_bench.py 2>&1 | grep -E " 16:| hw K" | head -8
timeout 600 python tools/dev/flag_ab.py 16 2>&1 | tail -6
