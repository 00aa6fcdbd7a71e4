timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "x3" 2>&1 | tail -5
X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 up128_512 2>&1 | grep mma | cut -c1-140
RES=1 X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 2>&1 | grep mma | cut -c1-140
KEEP_X3_HALO=b X3=1 timeout 300 python tools/bench_conv.py c128_256 2>&1 | grep mma | cut -c1-140
timeout 600 python tools/run_step.py x3 16 3 2>&1 | tail -1
