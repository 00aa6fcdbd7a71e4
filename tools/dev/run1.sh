python -m pytest tests -m gpu -x -q -k "streamed_sequence" 2>&1 | grep -v "it/s\]" | tail -60
python tools/dev/planref_ab.py
