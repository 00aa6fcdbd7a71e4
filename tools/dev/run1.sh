cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_facelib.py -x -q -m gpu -s -k "yolo" 2>&1 | grep -v "^$" | cut -c1-400 | tail -30
