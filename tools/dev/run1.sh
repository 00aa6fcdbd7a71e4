cd /root/repo
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "small_tile or partials or replica" 2>&1 | tail -4
python tools/dev/flag_ab.py 1 spec=0 nospec=16384 2>&1 | grep -v Warning | tail -8
python tools/dev/flag_ab.py 2 spec=0 nospec=16384 2>&1 | grep -v Warning | tail -5
python tools/dev/product_leg.py config3 2>&1 | grep "product_leg" | cut -c1-250
python tools/dev/product_leg.py config3 2>&1 | grep "process_image_sequence" | sed 's/.*process_image_sequence/pis/' | cut -c1-200
