cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "small_tile" 2>&1 | tail -3
python tools/dev/x3q_time.py 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/dev/flag_ab.py 1 new=0 co64=0x1000 2>&1 | tail -4
