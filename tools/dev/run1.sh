cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_x3" 2>&1 | tail -4
python tools/dev/lib_ab.py --b 1 --rounds 3 new=default old=comfyui-keep_amd/csrc/ab/lib_oldgemm.so 2>&1 | grep -v Warning | tail -8
python tools/dev/gemm_lat_bench.py 2>&1 | grep -v Warning | head -30
KEEP_HIP_LIB=comfyui-keep_amd/csrc/ab/lib_oldgemm.so python tools/dev/gemm_lat_bench.py 2>&1 | grep -v Warning | head -30
