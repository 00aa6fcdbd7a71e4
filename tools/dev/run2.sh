mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -rf -k "not asian" 2>&1 | grep -v "^  File\|^Extension" | tail -120 > gpurun_out/r2b/net.log
for p in fp32 x3; do
KEEP_DEBUG_SYNC=1 KEEP_AMD_PRECISION=$p timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k asian > gpurun_out/r2b/asian_$p.log 2>&1
tail -5 gpurun_out/r2b/asian_$p.log
done
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf 2>&1 | tail -40 > gpurun_out/r2b/kernels.log
tail -30 gpurun_out/r2b/kernels.log
