mkdir -p gpurun_out/r2f
for e in 0 6 7 8; do
  echo "=== KEEP_X3_EXP=$e" >> gpurun_out/r2f/ablate.log
  if [ $e = 0 ]; then X3=1 timeout 300 python tools/bench_conv.py c128_256 up128_512 2>&1 | grep mma >> gpurun_out/r2f/ablate.log
  else KEEP_X3_EXP=$e X3=1 timeout 300 python tools/bench_conv.py c128_256 up128_512 2>&1 | grep mma >> gpurun_out/r2f/ablate.log; fi
done
cat gpurun_out/r2f/ablate.log
