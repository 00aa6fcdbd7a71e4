cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "cfa" 2>&1 | tail -8
