cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -m gpu 2>&1 | tail -3
