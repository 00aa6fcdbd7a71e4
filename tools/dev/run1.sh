set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -8
timeout 600 python tools/dev/attn_flag_ab.py 1 new=0 old=16 2>&1 | tail -6
timeout 600 python tools/dev/attn_flag_ab.py 16 new=0 old=16 2>&1 | tail -6
