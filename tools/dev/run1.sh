python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "shapes_of_the_step or tensor2img" 2>&1 | grep -v "^$" | cut -c1-400 | tail -25
