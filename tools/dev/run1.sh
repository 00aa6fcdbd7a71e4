set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "small_tile_partials or halo or splitk" 2>&1 | tail -8
timeout 600 python tools/dev/flag_ab.py 1 new=0 old=0x2000 2>&1 | tail -6
