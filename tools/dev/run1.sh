python -m pytest tests/test_gpu_facelib.py -x -q -m gpu -s -k "mobile025 or dwconv or retinaface" 2>&1 | grep -v "^$" | cut -c1-300 | tail -25
python tools/dev/det_prof.py x3 mobile0.25 2>&1 | tail -3
python tools/dev/det_prof.py fp32 mobile0.25 2>&1 | tail -3
