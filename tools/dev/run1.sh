cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "small_tile or partials or replica or halo" 2>&1 | tail -6
python -m pytest tests/test_gpu_net.py -x -q -m gpu -k "cft or T3_vs_reference or batched_clips or config3 or hipgraph or two_stream" 2>&1 | tail -4
python tools/dev/lib_ab.py --b 1 --rounds 2 new=default old=comfyui-keep_amd/csrc/ab/lib_old.so 2>&1 | grep -v Warning | tail -5
python tools/dev/lib_ab.py --b 16 --rounds 2 new=default old=comfyui-keep_amd/csrc/ab/lib_old.so 2>&1 | grep -v Warning | tail -5
