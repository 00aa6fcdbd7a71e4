mkdir -p gpurun_out/r2e
for e in 0 1 2 3 4 5; do
  echo "=== KEEP_X3_EXP=$e" >> gpurun_out/r2e/ablate.log
  if [ $e = 0 ]; then X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 up128_512 2>&1 | grep mma >> gpurun_out/r2e/ablate.log
  else KEEP_X3_EXP=$e X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 up128_512 2>&1 | grep mma >> gpurun_out/r2e/ablate.log; fi
done
cat gpurun_out/r2e/ablate.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "x3" 2>&1 | tail -5
timeout 600 python tools/run_step.py x3 16 3 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2e/prof_x3 -o x3 -- python /root/repo/tools/run_step.py x3 16 2 > /root/repo/gpurun_out/r2e/prof.log 2>&1
cd /root/repo && python profiles/summarize_rocpd.py gpurun_out/r2e/prof_x3/x3_results.db 2 | head -24
