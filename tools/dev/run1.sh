cd /root/repo
python tools/dev/parse_prof.py census 2>&1 | grep -v amdgpu | cut -c1-200
