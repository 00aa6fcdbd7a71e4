set -x
cd /root/repo
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python tools/dev/flag_ab.py 1 new=0 old=0x2000 2>&1 | tail -4
timeout 600 python tools/dev/flag_ab.py 16 new=0 old=0x2000 2>&1 | tail -4
