for sv in 3; do
  echo "=== KEEP_X3P_SCHED=$sv"
  KEEP_X3P_SCHED=$sv X3=1 timeout 300 python tools/bench_conv.py c128_256 up128_512 2>&1 | grep mma | cut -c1-140
done
