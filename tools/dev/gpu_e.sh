python tools/dev/det_probe.py 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention or x3" 2>&1 | tail -5
