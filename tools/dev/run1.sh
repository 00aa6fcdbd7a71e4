cd /root/repo
for e in "A=1" "KEEP_X3_NO_TILE3=1"; do
  echo "== $e"
  env $e X3=1 python tools/bench_conv.py c128_64_1x1 2>&1 | grep -v amdgpu.ids | cut -c1-200
  env $e NOSTATS=1 X3=1 python tools/bench_conv.py c128_64_1x1 2>&1 | grep -v amdgpu.ids | cut -c1-200
  env $e DOWN=1 X3=1 python tools/bench_conv.py down64_512 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | grep -v "^$" | cut -c1-300 | tail -6
