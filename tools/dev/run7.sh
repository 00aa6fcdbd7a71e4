mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "x3" 2>&1 | tail -15
X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 up128_512 c256_64 2>&1 | grep mma | cut -c1-140
KEEP_NO_HALO_X3P=1 X3=1 timeout 300 python tools/bench_conv.py c128_256 2>&1 | grep mma | cut -c1-140
RES=1 X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 2>&1 | grep mma | cut -c1-140
timeout 600 python tools/run_step.py x3 16 3 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2g/prof_x3 -o x3 -- python /root/repo/tools/run_step.py x3 16 2 > /root/repo/gpurun_out/r2g/prof.log 2>&1
cd /root/repo && python profiles/summarize_rocpd.py gpurun_out/r2g/prof_x3/x3_results.db 2 | head -16
