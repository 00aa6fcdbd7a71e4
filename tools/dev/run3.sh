mkdir -p gpurun_out/r2c
KEEP_DEBUG_SYNC=1 timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -x -s -k asian 2>&1 | grep -v "^  File\|^Extension" > gpurun_out/r2c/asian_x3.log
grep -n "NON-FINITE" -B3 gpurun_out/r2c/asian_x3.log | head -40
tail -5 gpurun_out/r2c/asian_x3.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -m gpu -q -rf -k "not asian" 2>&1 | grep -v "^  File\|^Extension" | tail -30 > gpurun_out/r2c/all.log
tail -8 gpurun_out/r2c/all.log
