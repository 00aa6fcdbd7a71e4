set -x
cd /root/repo
timeout 600 python tools/dev/conv_split_bench.py 0x8 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python tools/dev/flag_ab.py 1 new=0 2>&1 | tail -3
