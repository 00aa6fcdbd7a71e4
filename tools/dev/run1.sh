cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_facelib.py -x -q -m gpu -s -k "retina" 2>&1 | grep -v "^$" | grep -v "max-abs" | cut -c1-300 | tail -8
