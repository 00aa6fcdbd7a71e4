cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" 2>&1 | tail -4
