python -m pytest tests/test_gpu_net.py -x -q -m gpu -s -k "pool" 2>&1 | grep -v "^$" | cut -c1-300 | tail -8
