# correctness of the ticket build: swap the library in place for the x3 kernel tests, then restore
cp comfyui-keep_amd/csrc/libkeep_hip.so /tmp/libkeep_hip.orig.so
cp comfyui-keep_amd/csrc/libkeep_dyn.so comfyui-keep_amd/csrc/libkeep_hip.so
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "x3" 2>&1 | tail -3
cp /tmp/libkeep_hip.orig.so comfyui-keep_amd/csrc/libkeep_hip.so
for i in 1 2; do
for L in c64_512 c128_256; do
X3=1 python tools/bench_conv.py $L 2>&1 | grep "True" | cut -c1-130
ABL_LIB=$PWD/comfyui-keep_amd/csrc/libkeep_dyn.so X3=1 python tools/bench_conv.py $L 2>&1 | grep "True" | cut -c1-130
done
done
