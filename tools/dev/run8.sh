for sv in 0 1 2; do
  echo "=== KEEP_X3P_SCHED=$sv"
  if [ $sv = 0 ]; then X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 up128_512 2>&1 | grep mma | cut -c1-140
  else KEEP_X3P_SCHED=$sv X3=1 timeout 300 python tools/bench_conv.py c128_256 c64_512 up128_512 2>&1 | grep mma | cut -c1-140; fi
done
KEEP_X3P_SCHED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "x3_halo" 2>&1 | tail -3
