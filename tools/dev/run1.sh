cd /root/repo
KEEP_HIP_LIB=comfyui-keep_amd/csrc/ab/lib_fused2.so python tools/dev/x3q_time.py 2>&1 | grep -v Warning | tail -4
KEEP_HIP_LIB=comfyui-keep_amd/csrc/ab/lib_dist3.so python tools/dev/x3q_time.py 2>&1 | grep -v Warning | tail -4
KEEP_HIP_LIB=comfyui-keep_amd/csrc/ab/lib_dist3.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "small_tile" 2>&1 | tail -2
