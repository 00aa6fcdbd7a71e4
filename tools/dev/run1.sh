cd /root/repo
timeout 900 python -m pytest tests/test_gpu_paste.py -x -q -m gpu 2>&1 | grep -v "^$" | cut -c1-300 | tail -12
