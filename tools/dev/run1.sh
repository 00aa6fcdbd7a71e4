cd /root/repo
timeout 900 python -m pytest tests/test_gpu_facelib.py -x -q -m gpu -k yolo 2>&1 | tail -5
