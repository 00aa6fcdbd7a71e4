cd /root/repo
python tools/dev/x3q_time.py 2>&1 | grep -v Warning | tail -5
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "small_tile or partials or replica" 2>&1 | tail -4
python tools/dev/lib_ab.py --b 1 --rounds 3 head=comfyui-keep_amd/csrc/ab/lib_head.so fused=comfyui-keep_amd/csrc/ab/lib_fused.so fused2=comfyui-keep_amd/csrc/ab/lib_fused2.so 2>&1 | grep -v Warning | tail -9
python tools/dev/lib_ab.py --b 16 --rounds 1 head=comfyui-keep_amd/csrc/ab/lib_head.so fused2=comfyui-keep_amd/csrc/ab/lib_fused2.so 2>&1 | grep -v Warning | tail -3
