cd /root/repo
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -x -q -m gpu -k "layernorm or code_prediction or cfa or kalman or T3_vs_reference" 2>&1 | tail -4
python tools/dev/lib_ab.py --b 1 --rounds 3 new=default old=comfyui-keep_amd/csrc/ab/lib_oldops.so 2>&1 | grep -v Warning | tail -8
