python -m pytest tests/test_gpu_net.py -x -q -m gpu -s 2>&1 | grep -v "^$" | cut -c1-1300 | tail -45
