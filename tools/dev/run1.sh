cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_x3_streaming" 2>&1 | grep -v "^$" | cut -c1-300 | tail -8
for e in "A=1" "KEEP_X3_NO_GEMM_STREAM=1" "ABL_LIB=$PWD/comfyui-keep_amd/csrc/libkeep_gx1.so"; do
  echo "== $e"
  env $e NOSTATS=1 X3=1 python tools/bench_conv.py lin256_1024 lin128 lin1024_128 c128_64_1x1 2>&1 | grep -v amdgpu.ids | cut -c1-200
  env $e NOSTATS=1 X3=1 ACT=gelu python tools/bench_conv.py lin256_1024 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
