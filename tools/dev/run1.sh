python -m pytest tests/test_gpu_net.py tests/test_gpu_facelib.py -x -q -m gpu -s -k "not full_forward" 2>&1 | grep -v "^$" | cut -c1-300 | tail -14
